"""bench.py's bookkeeping that does not need a GPU: the committed-profile lookups behind `roofline.traffic` and the rocprofv3
figures quoted next to the event figures."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_pmc_traffic_is_quoted_only_at_a_matching_vehicle_count():
    """A committed PMC summary belongs to the vehicle count of its own pass: quoted when this run's window mean is within
    15 %, refused (None + the reason) otherwise, and per configuration (no summary: None)."""
    b = _bench()
    for cfg, suffix in (('c3', ''), ('c2', '_c2'), ('c5', '_c5')):
        d = json.load(open(os.path.join(ROOT, 'profiles', '%s_pmc%s.json' % (b.PROFILE_TAGS[0], suffix))))
        v = float(d['mean_live_vehicles_per_env'])
        k = d['kernels']['env_step']
        got, src = b.pmc_traffic('env_step', cfg, v * 1.1)
        ff, fw, fsrc = b.fetch_calibration()            # the counters corrected by what the box reports on known 1-GiB streams
        assert abs(got - (ff * k['fetch_kb'] + fw * k['write_kb']) * 1024.0) < 1.0 and 'committed measurement' in src and 'whole episodes' in src
        assert 'FETCH_SIZE x' in src and fsrc.split()[0] in src
        got, why = b.pmc_traffic('env_step', cfg, v * 1.3)
        assert got is None and 'refused' in why and '15 %' in why
        got, why = b.pmc_traffic('env_step', cfg, v * 0.7)
        assert got is None and 'refused' in why
    assert b.pmc_traffic('env_step', 'q1', 600.0)[0] is None and b.pmc_traffic('env_step', None, 600.0)[0] is None


def test_fetch_calibration_is_the_committed_one_and_sane():
    """FETCH_SIZE / WRITE_SIZE are calibrated on known byte counts (tools/fetch_calib.hip); gfx950's rocprofv3
    reports half the bytes of a streaming read (MI355X_MICROARCH.md) and exact writes."""
    b = _bench()
    ff, fw, src = b.fetch_calibration()
    assert 'fetch_calibration.json' in src and 1.8 < ff < 2.2 and 0.9 < fw < 1.1
    d = json.load(open(os.path.join(ROOT, 'profiles', '%s_fetch_calibration.json' % b.PROFILE_TAGS[0])))
    assert d['known_bytes'] == 1 << 30 and set(d['fetch']) == {'read_dword', 'read_f4'} and 'write_f4_nt' in d['write']


def test_iql_algorithmic_flops_count_the_block_diagonal_first_layer():
    """bench.py:iql_algorithmic_flops (the q1 roofline): an inner large_grid agent (30 wave + 6 wait inputs, 5 actions) costs
    2 x 14 592 MACs forward + 24 896 backward per row; structural zeros of W1 are not counted."""
    b = _bench()

    class Lay: n_fc0, ft, H1, H2 = 128, 32, 160, 64

    class M: layout, n_wave_ls, n_w_ls, n_a_ls = Lay, [30], [6], [5]
    fl = b.iql_algorithmic_flops(M, 10)
    fwd = 30 * 128 + 6 * 32 + 160 * 64 + 64 * 5
    bwd = 64 * 5 + 64 + 2 * 160 * 64 + 30 * 128 + 6 * 32
    assert fl['iql_act'] == 2.0 * fwd * 10 and fl['iql_grad'] == 2.0 * (2 * fwd + bwd) * 10


def test_rocprofv3_averages_come_from_the_configuration_s_own_trace():
    b = _bench()
    us3, src3 = b.rocprof_avg_us('step_kernel', 'c3')
    us2, src2 = b.rocprof_avg_us('step_kernel', 'c2')
    us5, _ = b.rocprof_avg_us('step_kernel', 'c5')
    assert src3.endswith('_kernel_stats.csv') and src2.endswith('_kernel_stats_c2.csv')
    assert 60 < us3 < 110 and 30 < us2 < 70 and 35 < us5 < 80 and us2 < us3          # 256 threads / 1024 instances vs 1024 threads / 256 instances
    fw, _ = b.rocprof_avg_us('policy_fwd_', 'c3')
    assert 85 < fw < 110
    assert b.rocprof_avg_us('step_kernel', None) == (None, None) and b.rocprof_avg_us('no_such_kernel', 'c3') == (None, None)


def test_presets_name_the_baseline_configurations():
    b = _bench()
    assert b.preset_name('large_grid', 'ma2c', 'lstm', 1024) == 'c3' and b.preset_name('large_grid', 'ia2c', 'fc', 256) == 'c2'
    assert b.preset_name('real_net', 'ma2c', 'lstm', 512) == 'c5' and b.preset_name('large_grid', 'iqld', 'dqn', 1024) == 'q1'
    assert b.preset_name('large_grid', 'ma2c', 'lstm', 64) is None


def test_default_window_is_two_whole_episodes_after_whole_warm_up_episodes():
    """>= 2 full episodes timed after >= 1 warm-up episode.  An episode is 720 control steps; an iteration is the
    agent's n_step control steps (config/*.ini batch_size: 120 large_grid A2C, 40 Monaco, 20 IQL); the warm-up is whole
    episodes and at least 10 iterations (a cold device needs that long to reach its clocks)."""
    b = _bench()
    for (scenario, agent), (ipe, warm) in {('large_grid', 'ma2c'): (6, 12), ('large_grid', 'ia2c'): (6, 12), ('real_net', 'ma2c'): (18, 18),
                                           ('large_grid', 'iqld'): (36, 36), ('large_grid', 'iqll'): (36, 36)}.items():
        assert b.iterations_per_episode(scenario, agent) == ipe
        assert b.warmup_iterations(ipe) == warm and warm % ipe == 0 and warm >= 10
    # the package's defaults carry the reference's large_grid values (config_ma2c_large.ini / config_iqld_large.ini [MODEL_CONFIG])
    from deeprl_signal_control_amd.agents import A2C_DEFAULTS
    from deeprl_signal_control_amd.iql import IQL_DEFAULTS
    assert 720 // A2C_DEFAULTS['batch_size'] == b.iterations_per_episode('large_grid', 'ma2c')
    assert 720 // IQL_DEFAULTS['batch_size'] == b.iterations_per_episode('large_grid', 'iqld')
