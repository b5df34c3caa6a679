"""The compile-time table dimensions of the specialised simulator kernel against the scenario compiler (CPU)."""
import os
import re

import numpy as np

from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net


def _reachable_prefix(scn):
    """Lanes a route can ever put a vehicle on (tsc_env_create: route entry lane -> mv_next chain); 1 + the highest one."""
    nu = 0
    nxt = np.asarray(scn.mv_next).reshape(scn.n_lane, scn.n_route)
    for r in range(scn.n_route):
        l, hops = int(scn.route_entry_lane[r]), 0
        while 0 <= l < scn.n_lane and hops <= 2 * scn.n_lane:
            nu = max(nu, l + 1)
            nx = int(nxt[l, r])
            if nx < -1 and scn.lane_sib is not None and scn.lane_sib[l] >= 0 and nxt[scn.lane_sib[l], r] >= -1:
                nx = int(scn.lane_sib[l])                       # rule 10: the vehicle moves over to the lane that serves it
            l = nx; hops += 1
    return nu


def test_specialised_kernel_dimensions_match_the_scenarios():
    """step_kernel's compile-time table dimensions (kSpec in csrc/tsc_env.hip) are matched against the scenario at create time
    and fall back to the launch-time kernel when they differ: pin them to what the scenario compiler produces, so that a
    change of the compiler does not quietly un-specialise the benchmarked path."""
    src = open(os.path.join(os.path.dirname(__file__), '..', 'deeprl_signal_control_amd', 'csrc', 'tsc_env.hip')).read()
    rows = re.search(r'constexpr SpecDims kSpec\[3\] = \{(.*?)\};', src, re.S).group(1)
    spec = [tuple(int(v) for v in r.split(',')) for r in re.findall(r'\{([^{}]*)\}', rows)]
    assert len(spec) == 3 and spec[0] == (0,) * 12
    for k, scn in ((1, build_large_grid('ma2c', lane_change=False)), (1, build_large_grid('ma2c', lane_change=True)), (2, build_real_net('ma2c'))):
        NLP, NLA, NU, NR, A, KMAX, PMAX, LMAX, NBR, ctrl, yellow, teleport = spec[k]
        assert NLP == (scn.n_lane + 63) // 64 * 64
        # the scenario's live lanes (81 on large_grid, 83 with lane changing, 113 on Monaco) are a prefix of the instantiation's
        assert _reachable_prefix(scn) <= NU <= scn.n_lane and NU - _reachable_prefix(scn) < 8 and NLA == (NU + 63) // 64 * 64
        assert NR == scn.n_route and A == scn.n_agent
        assert (PMAX, KMAX) == tuple(scn.green_tab.shape[1:]) and LMAX == scn.agent_lanes.shape[1]
        assert NBR == max(1, max(len(n) for n in scn.neighbors))
        assert (ctrl, yellow, teleport) == (scn.control_interval_sec, scn.yellow_interval_sec, scn.teleport_sec)
    ia = build_large_grid('ia2c')                          # same tables, narrower observations: same instantiation
    assert (ia.n_lane, ia.n_route, ia.n_agent) == (180, 12, 25)


def test_contracted_chains_drop_no_right_of_way_relation():
    """contract_chains refuses to merge a lane whose movements yield or have priority; on the reference's Monaco
    routes no such movement exists (every unsignalised merge is a zipper), so the contraction (160 -> 113 live lanes) loses
    nothing."""
    from deeprl_signal_control_amd.scenario import build_real_net
    scn = build_real_net('ma2c')
    assert int((scn.mv_yield >= 0).sum()) == 0 and int((scn.mv_prio != 0).sum()) == 0
    assert scn.n_lane == 158


def test_capacity_limits_are_refused_not_truncated():
    """a scenario that exceeds the fixed device capacities raises with the remedy."""
    import pytest
    from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net, build_small_grid
    for scn in (build_large_grid('ma2c'), build_large_grid('ma2c', init_density=0.2), build_real_net('ma2c'), build_small_grid('greedy')):
        scn.streams_ready()
        scn.check_limits()                                 # the reference's scenarios fit
    scn = build_large_grid('ma2c')
    scn.lane_len = scn.lane_len.copy()
    scn.lane_len[0] = 400.0
    with pytest.raises(ValueError, match='split them'):
        scn.check_limits()


def test_foe_tables_of_the_junction_experiment():
    """Scenario.link_foes (MICROSIM_SPEC.md "Junction interiors": data of a round-4 experiment on the CPU oracle, not consumed by the
    device library).  Four-leg grid junctions: chord crossing over netconvert's link order -- symmetric, no link is its own
    foe, links of one approach are never foes, a right turn only meets the two streams that join its target, the opposing
    through streams do not conflict, a permissive left crosses the opposing through.  Monaco: the junctions' own
    right-of-way matrices (<request foes> of most.net.xml, compiled by tools/compile_real_net.py) -- symmetric as well."""
    from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net, four_leg_foes
    f = four_leg_foes()
    name = [leg + turn for leg in 'NESW' for turn in 'RSL']
    foes = {name[k]: {name[j] for j in range(12) if (int(f[k]) >> j) & 1} for k in range(12)}
    for k in range(12):
        assert not (int(f[k]) >> k) & 1
        for j in range(12):
            assert ((int(f[k]) >> j) & 1) == ((int(f[j]) >> k) & 1)
            if j // 3 == k // 3:
                assert not (int(f[k]) >> j) & 1
    assert foes['NR'] == {'ES', 'SL'} and 'SS' not in foes['NS'] and 'SS' in foes['NL'] and 'ES' in foes['NS']
    lg = build_large_grid('ma2c')
    assert lg.link_foes.shape == (25, 12) and (lg.link_foes == f[None, :]).all()
    rn = build_real_net('ma2c')
    assert rn.link_foes.shape == (rn.n_agent, rn.green_tab.shape[2]) and (rn.link_foes != 0).sum() > 150
    for a in range(rn.n_agent):
        for k in range(int(rn.agent_nlink[a])):
            for j in range(int(rn.agent_nlink[a])):
                assert ((int(rn.link_foes[a, k]) >> j) & 1) == ((int(rn.link_foes[a, j]) >> k) & 1), (a, k, j)
            assert not (int(rn.link_foes[a, k]) >> k) & 1


def test_greedy_controller_tables_restate_the_three_reference_controllers():
    """Scenario.greedy_controller_tables (what tsc_env_set_greedy uploads and trainer.greedy_actions walks on the host):
    large_grid = LargeGridController's hard-coded lane pairs (envs/large_grid_env.py:56-60, restated by the oracle's
    greedy_large_grid) incl. first-argmax ties and the float64 non-tie 0.2 + 0.4 > 0.6; small_grid = STATE_PHASE_MAP
    (envs/small_grid_env.py:29-30,51-55); Monaco = the 'G' links of every phase in link order, every lane once
    (envs/real_net_env.py:90-111; its recorded answers: tests/test_real_net.py)."""
    from deeprl_signal_control_amd.scenario import build_small_grid
    from deeprl_signal_control_amd.trainer import greedy_actions
    from oracle.env_oracle import greedy_large_grid
    rng = np.random.RandomState(3)
    lg = build_large_grid('greedy')
    n_cand, term, action = lg.greedy_controller_tables()
    assert n_cand.tolist() == [5] * 25 and term[0].tolist() == [[0, 3], [2, 5], [1, 4], [1, 2], [4, 5]] and action[0].tolist() == [0, 1, 2, 3, 4]
    ob = np.clip(rng.randint(0, 12, (40, 25, 6)) / 5.0, 0, 2.0)
    ob[0, 0] = [0.2, 0.0, 0.6, 0.4, 0.0, 0.0]
    want = np.array([[greedy_large_grid(ob[e, a]) for a in range(25)] for e in range(40)])
    np.testing.assert_array_equal(greedy_actions(lg, ob), want)
    assert want[0, 0] == 0
    sg = build_small_grid('greedy')
    n_cand, term, action = sg.greedy_controller_tables()
    spm = sg.extra['state_phase_map']
    for a, n in enumerate(sg.node_names):
        assert n_cand[a] == len(spm[n]) and action[a, :n_cand[a]].tolist() == list(spm[n])
        assert term[a, :n_cand[a], 0].tolist() == list(range(len(spm[n]))) and (term[a, :, 1:] == -1).all()
    w = rng.rand(10, sg.n_agent, 3)
    for e in range(10):
        for a, n in enumerate(sg.node_names):
            assert greedy_actions(sg, w)[e, a] == spm[n][int(np.argmax(w[e, a, :len(spm[n])]))]
    rn = build_real_net('greedy')
    n_cand, term, action = rn.greedy_controller_tables()
    assert n_cand.tolist() == [int(x) for x in rn.agent_nphase] and (action[0, :n_cand[0]] == np.arange(n_cand[0])).all()
    for a in range(rn.n_agent):
        for c in range(int(n_cand[a])):
            t = [int(x) for x in term[a, c] if x >= 0]
            assert len(t) == len(set(t)) and all(x < int(rn.agent_nlane[a]) for x in t)      # every lane once, own lanes only
