"""Host-side configuration surface (main.py:21-48,51-79,82-155 of the reference) without a GPU: INI sections of the
reference's config files (restated inline: /root/reference is not read), CLI flags, on-disk helpers."""
import configparser
import os

import numpy as np
import pandas as pd
import pytest

from deeprl_signal_control_amd import main as cli
from deeprl_signal_control_amd.agents import A2C_DEFAULTS, Scheduler, coerce_config, replica_sample_seed
from deeprl_signal_control_amd.env import scenario_from_config
from deeprl_signal_control_amd.iql import IQL_DEFAULTS

# the key set of config/config_ma2c_large.ini ([ENV_CONFIG], [MODEL_CONFIG], [TRAIN_CONFIG])
INI = """
[MODEL_CONFIG]
rmsp_alpha = 0.99
rmsp_epsilon = 1e-5
max_grad_norm = 40
gamma = 0.99
lr_init = 5e-4
lr_decay = constant
entropy_coef_init = 0.01
entropy_coef_min = 0.01
entropy_decay = constant
entropy_ratio = 0.5
value_coef = 0.5
num_lstm = 64
num_fw = 128
num_ft = 32
num_fp = 64
batch_size = 120
reward_norm = 2000.0
reward_clip = 2.0

[TRAIN_CONFIG]
total_step = 1e6
test_interval = 2e6
log_interval = 1e4

[ENV_CONFIG]
clip_wave = 2.0
clip_wait = 2.0
control_interval_sec = 5
agent = ma2c
coop_gamma = 0.9
data_path = ./large_grid/data/
episode_length_sec = 3600
norm_wave = 5.0
norm_wait = 100.0
coef_wait = 0.2
peak_flow1 = 1100
peak_flow2 = 925
init_density = 0
objective = hybrid
scenario = large_grid
seed = 12
test_seeds = 10000,20000
yellow_interval_sec = 2
"""


def _config():
    c = configparser.ConfigParser()
    c.read_string(INI)
    return c


def test_env_section_builds_the_scenario():
    scn, seed, test_seeds = scenario_from_config(_config()['ENV_CONFIG'])
    assert (scn.name, scn.agent, seed, test_seeds) == ('large_grid', 'ma2c', 12, (10000, 20000))
    assert (scn.n_agent, scn.s_max, scn.control_interval_sec, scn.yellow_interval_sec) == (25, 52, 5, 2)
    assert (scn.norm_wave, scn.norm_wait, scn.clip_wave, scn.coef_wait, scn.coop_gamma) == (5.0, 100.0, 2.0, 0.2, 0.9)
    # the demand follows the peak flows of the section (large_grid/data/build_file.py:284-324)
    base = np.asarray(scn.flows)[:, 2].sum()
    cfg = _config()
    cfg['ENV_CONFIG']['peak_flow1'] = '2200'
    assert np.asarray(scenario_from_config(cfg['ENV_CONFIG'])[0].flows)[:, 2].sum() > base
    cfg['ENV_CONFIG']['init_density'] = '0.3'              # initial traffic (large_grid/data/build_file.py:223-266): 120 more streams
    scn_d = scenario_from_config(cfg['ENV_CONFIG'])[0]
    assert scn_d.n_stream == 12 + 120 and scn_d.extra['car_num'] == 9 and scn_d.n_route == 20


def test_model_section_is_typed_like_the_reference_getters():
    cfg = coerce_config(_config()['MODEL_CONFIG'], A2C_DEFAULTS)
    assert cfg['lr_init'] == 5e-4 and cfg['batch_size'] == 120 and isinstance(cfg['batch_size'], int)
    assert cfg['lr_decay'] == 'constant' and cfg['reward_norm'] == 2000.0 and cfg['max_grad_norm'] == 40.0
    q = coerce_config({'LR_INIT': '1e-4', 'lr_decay': 'linear', 'LR_MIN': '1e-5', 'buffer_size': '1e3', 'batch_size': '20',
                       'epsilon_init': '1.0', 'epsilon_min': '0.01', 'epsilon_decay': 'linear', 'epsilon_ratio': '0.5'}, IQL_DEFAULTS)
    assert q['lr_init'] == 1e-4 and q['lr_min'] == 1e-5 and q['lr_decay'] == 'linear' and q['batch_size'] == 20
    s = Scheduler(q['lr_init'], q['lr_min'], 1000, decay=q['lr_decay'])          # agents/utils.py:268-281 on the coerced values
    assert s.get(500) == pytest.approx(5e-5) and s.get(500) == pytest.approx(1e-5) and s.get(10) == pytest.approx(1e-5)


def test_cli_flags_and_disk_helpers(tmp_path):
    a = cli.parse_args(['--base-dir', str(tmp_path), 'train', '--config-dir', 'x.ini', '--test-mode', 'in_train_test', '--envs', '64'])
    assert (a.option, a.config_dir, a.test_mode, a.envs) == ('train', 'x.ini', 'in_train_test', 64)
    e = cli.parse_args(['--base-dir', str(tmp_path), 'evaluate', '--agents', 'ma2c,greedy', '--evaluation-seeds', '1,2'])
    assert (e.option, e.agents, e.evaluation_seeds, e.evaluation_policy_type) == ('evaluate', 'ma2c,greedy', '1,2', 'default')
    assert cli.parse_args(['evaluate']).evaluation_seeds == ','.join(str(i) for i in range(10000, 100001, 10000))   # main.py:44-46
    assert [cli.init_test_flag(m) for m in ('no_test', 'in_train_test', 'after_train_test', 'all_test')] == \
        [(False, False), (True, False), (False, True), (True, True)]                                              # utils.py:51-60
    dirs = cli.init_dir(str(tmp_path / 'run'))
    assert sorted(dirs) == ['data', 'log', 'model'] and all(os.path.isdir(d) for d in dirs.values())
    open(dirs['data'] + 'b.ini', 'w').close(); open(dirs['data'] + 'a.ini', 'w').close()
    assert cli.find_file(dirs['data'].rstrip('/')).endswith('a.ini')
    rows = [dict(agent='ma2c', step=120, test_id=-1, avg_reward=-1.5, std_reward=0.5), dict(agent='ma2c', step=240, test_id=-1, avg_reward=-1.0, std_reward=0.2)]
    cli.write_reward_csv(rows, dirs['data'] + 'train_reward.csv')
    df = pd.read_csv(dirs['data'] + 'train_reward.csv', index_col=0)
    assert list(df.columns) == ['agent', 'avg_reward', 'std_reward', 'step', 'test_id'] and len(df) == 2             # utils.py:299-308


def test_replica_sampling_streams_are_distinct():
    seeds = {replica_sample_seed(12, r, b) for r in range(8) for b in range(2)}
    assert len(seeds) == 16 and replica_sample_seed(12) == 12


def test_checkpoint_seed_slot_keeps_its_meaning_per_format():
    """`counters` of a checkpoint hold the BASE seed (format 2; round 3 wrote the same without the marker,
    rounds 1-2 wrote rank 0's stream seed, which equals the base seed): every rank / replica re-derives its action stream,
    whatever the file's format entry says -- a marker-less file must not put all ranks on rank 0's stream."""
    from deeprl_signal_control_amd.agents import CKPT_FORMAT, replica_sample_seed, resume_sample_seed
    assert CKPT_FORMAT >= 2
    base, s0 = resume_sample_seed(7, CKPT_FORMAT, rank=0, replica=0)
    assert (base, s0) == (7, 7)                                  # rank 0 / replica 0 keeps the base stream
    base, s3 = resume_sample_seed(7, CKPT_FORMAT, rank=3, replica=1)
    assert base == 7 and s3 == replica_sample_seed(7, 3, 1) != 7
    assert resume_sample_seed(123456789, None, rank=3, replica=1) == (123456789, replica_sample_seed(123456789, 3, 1))
    assert resume_sample_seed(123456789, None, rank=0, replica=0) == (123456789, 123456789)


def test_initial_traffic_draw_handles_streams_with_different_candidate_counts():
    """draw_stream_routes: one vectorised RandomState.choice when every drawn stream has the same number of candidate sinks (the
    reference's initial traffic: the legacy stream of 120 scalar draws), a per-stream draw otherwise; candidates come from the
    stream table, not from scenario extras."""
    import numpy as np
    from deeprl_signal_control_amd.scenario import build_large_grid, draw_stream_routes
    scn = build_large_grid('ma2c', init_density=0.2, sort_lanes=False)
    m2 = np.nonzero(np.asarray(scn.stream_mode) == 2)[0]
    r = draw_stream_routes(scn, 12)
    rs = np.random.RandomState(12)
    cand = scn.stream_choice[m2, 0, :, 0]
    want = [int(cand[j, int(rs.choice(int((cand[j] >= 0).sum())))]) for j in range(len(m2))]      # 120 scalar draws
    assert [int(x) for x in r[m2]] == want
    scn.extra.pop('sink_routes', None)                          # not needed any more
    assert np.array_equal(draw_stream_routes(scn, 12), r)
    scn.stream_choice = scn.stream_choice.copy()
    scn.stream_choice[m2[0], 0, -1, 0] = -1                     # one stream with a candidate less: the per-stream branch
    r2 = draw_stream_routes(scn, 12)
    assert r2[m2[0]] in [int(x) for x in scn.stream_choice[m2[0], 0, :, 0] if x >= 0]
