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


# ---- the forward edge kernels' schedule (edge_mfma.hip: edge_active_waves + the XCD-aware partition of edge_body, and the role split
# of edge_x2h_dual_kernel), restated: every item exactly once whatever the list length, and short lists at one wave per SIMD ------------
MIN_WAVES = 2


def active_waves(n_items, n_wg, waves=8):
    per_wg = -(-n_items // max(n_wg, 1))
    return min(waves, max(min(MIN_WAVES, waves), per_wg))


WAVE_MAJOR = True      # CBGX_EDGE_WAVE_MAJOR: item o of a round -> (wave o // slots, workgroup o % slots)


def forward_schedule(n_items, n_wg, waves=8, wave_major=WAVE_MAJOR):
    """-> {(wg, wave): [items in processing order]} of edge_body"""
    wv = active_waves(n_items, n_wg, waves)
    out = {}
    for wg in range(n_wg):
        if n_wg % 8 == 0:
            per_xcd = (((n_items + 7) >> 3) + wv - 1) // wv * wv
            first = (wg & 7) * per_xcd + (wg >> 3) * (1 if wave_major else wv)
            if first >= min(n_items, ((wg & 7) + 1) * per_xcd):
                continue
        elif wg * (1 if wave_major else wv) >= n_items:
            continue
        for wave in range(waves):
            if wave >= wv:
                continue
            if n_wg % 8 == 0:
                xcd, slot, slots = wg & 7, wg >> 3, n_wg >> 3
                off = wave * slots + slot if wave_major else slot * wv + wave
                i_begin, i_end, i_step = xcd * per_xcd + off, min(n_items, (xcd + 1) * per_xcd), slots * wv
            else:
                i_begin, i_end, i_step = (wave * n_wg + wg if wave_major else wg * wv + wave), n_items, n_wg * wv
            if i_begin < i_end:
                out[(wg, wave)] = list(range(i_begin, i_end, i_step))
    return out


def launcher_grid(n_nodes, dual=False):
    grid = min(256, -(-n_nodes // MIN_WAVES) + (1 if dual else 0))
    if dual:
        return max(grid, 2)
    return grid & ~7 if grid >= 64 else grid


@pytest.mark.parametrize("n_wg", [1, 2, 9, 56, 57, 64, 112, 176, 256])
def test_forward_schedule_covers_every_item_once(n_wg):
    for n in [1, 5, 8, 9, 25, 250, 445, 1024, 1230, 3170, 4404, 16506, 99543]:
        for wm in (True, False):
            sched = forward_schedule(n, n_wg, wave_major=wm)
            assert sorted(i for v in sched.values() for i in v) == list(range(n)), (n, n_wg, wm)


def test_a_partial_last_round_spreads_over_the_workgroups():
    # ten graphs: 4450 nodes, 2.17 per wave; the protein-only role alone, 3200 nodes on 176 workgroups = 2.27 per wave: the items of
    # the third round go to wave 0 (then wave 1, ...) of every workgroup, not to all eight waves of the first few
    for n, n_wg in ((4450, 256), (3200, 176)):
        sched = forward_schedule(n, n_wg)
        third = [(wg, wave) for (wg, wave), v in sched.items() if len(v) == 3]
        assert third and max(wave for _, wave in third) <= 2
        per_wg = {}
        for wg, _ in third:
            per_wg[wg] = per_wg.get(wg, 0) + 1
        assert max(per_wg.values()) <= 3                              # at most one per SIMD
        old = forward_schedule(n, n_wg, wave_major=False)
        third_old = [(wg, wave) for (wg, wave), v in old.items() if len(v) == 3]
        assert len(third_old) == len(third) and max(wave for _, wave in third_old) == 7   # (the mapping this replaces)


def test_short_lists_run_one_wave_per_simd():
    # one graph (445 nodes), the movable atoms of a 10-graph batch (250 of 4450 nodes), a 1-graph h2x block (25 of 445): at most four
    # waves of a workgroup hold items -- waves 0..3, one per SIMD -- and every wave holds one item
    for n_items, n_nodes in ((445, 445), (250, 4450), (25, 445), (1000, 173558)):
        sched = forward_schedule(n_items, launcher_grid(n_nodes))
        assert max(wave for _, wave in sched) <= 3 and max(len(v) for v in sched.values()) == 1, (n_items, n_nodes)
        if n_nodes <= 500:      # one graph: two or three waves per workgroup (the grid is rounded down to a multiple of 8)
            assert max(wave for _, wave in sched) <= 2
    # a long list uses all eight waves of all 256 workgroups, balanced to one item
    sched = forward_schedule(173558, 256)
    lens = [len(v) for v in sched.values()]
    assert len(sched) == 2048 and max(lens) - min(lens) <= 1
    # in between (a cached layer of a 10-graph batch: ~1500 of 4450 nodes): six waves per workgroup, one item each
    sched = forward_schedule(1500, 256)
    assert max(wave for _, wave in sched) == 5 and max(len(v) for v in sched.values()) == 1


def dual_roles(c_pp, c_gen, n_wg, waves=8, gen_cost=1.15):
    """edge_x2h_dual_kernel's role split -> (workgroups of the protein-only role, workgroups of the general role)"""
    wv = active_waves(c_pp + c_gen, n_wg - 1 if n_wg > 1 else 1, waves)
    need_pp, need_gen = -(-c_pp // wv), -(-c_gen // wv)
    if c_gen == 0:
        n_pp = n_wg
    elif c_pp == 0:
        n_pp = 0
    elif need_pp + need_gen <= n_wg:
        n_pp = need_pp
    else:
        share = c_pp / (c_pp + gen_cost * c_gen)
        unit = 8 if n_wg >= 64 else 1
        n_pp = int(share * (n_wg // unit) + 0.5) * unit
        n_pp = max(unit, min(n_wg - unit, n_pp))
        if n_wg < 2:
            n_pp = 0
    return n_pp, n_wg - n_pp


def test_dual_roles_cover_both_lists():
    for c_pp, c_gen in [(320, 125), (0, 445), (445, 0), (3200, 1250), (1, 1), (7, 300), (125000, 48558), (900, 100)]:
        n_nodes = c_pp + c_gen
        grid = launcher_grid(n_nodes, dual=True)
        n_pp, n_gen = dual_roles(c_pp, c_gen, grid)
        assert n_pp + n_gen == grid and (n_pp > 0 or c_pp == 0) and (n_gen > 0 or c_gen == 0), (c_pp, c_gen, grid)
        for c, n in ((c_pp, n_pp), (c_gen, n_gen)):
            if c:
                sched = forward_schedule(c, n)
                assert sorted(i for v in sched.values() for i in v) == list(range(c))
                if n_nodes <= 1000:          # small input: one node per wave, one wave per SIMD, in both roles
                    assert max(len(v) for v in sched.values()) == 1 and max(wave for _, wave in sched) <= 3, (c_pp, c_gen)
