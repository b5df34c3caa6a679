"""The reference's command line (main.py:21-48,82-230) on the MI355X path: `train` writes DIR/{log,data,model} with the
config copy, train_reward.csv and checkpoint-<step>; `evaluate` reads DIR/<agent>/{data,model} and writes
DIR/eva_data/<scenario>_<agent>_{control,traffic,trip}.csv for every evaluation seed."""
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

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
num_fw = 128
num_ft = 32
num_lstm = 64
num_fp = 64
batch_size = 20
reward_norm = 2000.0
reward_clip = 2.0

[TRAIN_CONFIG]
total_step = 120
test_interval = 60
log_interval = 60

[ENV_CONFIG]
clip_wave = 2.0
clip_wait = 2.0
control_interval_sec = 5
agent = %(agent)s
coop_gamma = 0.9
data_path = ./large_grid/data/
episode_length_sec = 300
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
IQL_MODEL = """
[MODEL_CONFIG]
max_grad_norm = 40
gamma = 0.99
lr_init = 1e-4
lr_decay = constant
epsilon_init = 1.0
epsilon_min = 0.01
epsilon_decay = linear
epsilon_ratio = 0.5
num_fc = 128
num_h = 64
batch_size = 20
buffer_size = 1000
reward_norm = 3000.0
reward_clip = 2.0
"""


@pytest.mark.parametrize('agent', ['ma2c', 'iqld'])
def test_train_then_evaluate(agent, tmp_path):
    import pandas as pd
    from deeprl_signal_control_amd import main as cli
    ini = INI % {'agent': agent}
    if agent == 'iqld':
        ini = IQL_MODEL + ini[ini.index('[TRAIN_CONFIG]'):]
    cfg = tmp_path / ('config_%s.ini' % agent)
    cfg.write_text(ini)
    base = str(tmp_path / 'exp')
    rows = cli.main(['--base-dir', base + '/' + agent, 'train', '--config-dir', str(cfg), '--test-mode', 'in_train_test', '--envs', '4'])
    assert os.path.exists(base + '/%s/data/config_%s.ini' % (agent, agent))
    df = pd.read_csv(base + '/%s/data/train_reward.csv' % agent, index_col=0)
    assert list(df.columns) == ['agent', 'avg_reward', 'std_reward', 'step', 'test_id'] and len(df) == len(rows)
    train_rows = df[df.test_id == -1]
    assert list(train_rows.step) == [60, 120] and (train_rows.avg_reward < 0).all() and set(df.agent) == {agent}
    assert sorted(df[df.test_id >= 0].test_id.unique()) == [0, 1]                # the test block ran on both test seeds
    assert os.path.exists(base + '/%s/model/checkpoint-120.npz' % agent)
    # evaluate the trained agent and the greedy baseline on three seeds
    os.makedirs(base + '/greedy/data')
    shutil.copy(str(cfg), base + '/greedy/data/')
    out = cli.main(['--base-dir', base, 'evaluate', '--agents', '%s,greedy' % agent, '--evaluation-seeds', '10000,20000,30000'])
    for name in (agent, 'greedy'):
        mean, std = out[name]
        assert mean.shape == (3,) and (mean < 0).all()
        c = pd.read_csv(base + '/eva_data/large_grid_%s_control.csv' % name, index_col=0)
        assert sorted(c.episode.unique()) == [1, 2, 3] and len(c) == 3 * 60
        for e in range(3):                                                        # the control log holds the global reward
            assert abs(c[c.episode == e + 1].reward.mean() - mean[e]) < 1e-9
        t = pd.read_csv(base + '/eva_data/large_grid_%s_traffic.csv' % name, index_col=0)
        assert len(t) == 3 * 300 and (t.number_total_car >= 0).all()
        tr = pd.read_csv(base + '/eva_data/large_grid_%s_trip.csv' % name, index_col=0)
        # (an untrained argmax Q policy holds one phase: few vehicles get through in 300 s)
        assert len(tr) > (0 if name == 'iqld' else 50) and set(tr.episode.unique()) <= {1, 2, 3}
    g = pd.read_csv(base + '/eva_data/large_grid_greedy_control.csv', index_col=0)
    assert len(set(g[g.episode == 1].reward.round(6))) > 3                        # seeds differ, traffic is live
