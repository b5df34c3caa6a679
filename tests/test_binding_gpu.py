"""The reference-shaped binding: ``TrafficEnv`` + ``MA2C`` / ``IA2C`` (E = 1, lists of ndarrays) driven
by a line-for-line restatement of the reference's own loop (utils.py:142-193 ``Trainer.explore``, :255-308
``Trainer.run``), the config-section constructors ``main.py:51-127`` uses, the greedy controllers on device tensors,
and the multi-GPU plumbing of ``VecA2C.backward`` on the HIP model (1-rank RCCL group + two half-batch handles)."""
import configparser
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

INI = """
[MODEL_CONFIG]
rmsp_alpha = 0.99
rmsp_epsilon = 1e-5
max_grad_norm = 40
gamma = 0.99
lr_init = 5e-4
lr_decay = linear
LR_MIN = 1e-5
entropy_coef_init = 0.01
entropy_coef_min = 0.01
entropy_decay = constant
entropy_ratio = 0.5
value_coef = 0.5
num_fw = 128
num_ft = 32
num_lstm = 64
num_fp = 64
batch_size = 40
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
episode_length_sec = 600
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
    cfg = configparser.ConfigParser()
    cfg.read_string(INI)
    return cfg


def _choice(model, pi, agent_idx):
    """np.random.choice(np.arange(len(pi)), p=pi) (utils.py:155-157) on the library's documented uniform, so the
    reference loop and the batched trainer draw the same actions."""
    from oracle.nets_oracle import choice_from_uniform, sample_uniform
    return choice_from_uniform(pi, sample_uniform(model.vec.sample_seed, model.vec.sample_step, agent_idx))


def _reference_run(env, model, n_episode):
    """utils.py:255-308 + :142-193 for the a2c agents, statement by statement (logging / TF summaries dropped)."""
    n_step = model.n_step
    assert env.T % n_step == 0
    data = []
    for _ in range(n_episode):
        env.train_mode = True
        ob = env.reset()
        done = True                                        # pre-decision done: resets the LSTM states
        model.reset()
        rewards = []
        while True:
            # ---- explore (utils.py:142-193)
            for _ in range(n_step):
                policy, value = model.forward(ob, done)
                if env.agent == 'ma2c':
                    env.update_fingerprint(policy)
                action = [_choice(model, pi, a) for a, pi in enumerate(policy)]
                model.vec.sample_step += 1
                next_ob, reward, done, global_reward = env.step(action)
                rewards.append(global_reward)
                model.add_transition(ob, action, reward, value, done)
                if done:
                    break
                ob = next_ob
            R = [0] * model.n_agent if done else model.forward(ob, False, 'v')
            model.backward(R, None, 0)
            if done:
                env.terminate()
                break
        data.append((np.mean(rewards), np.std(rewards)))
    return data


def test_reference_loop_on_the_binding_equals_vec_trainer():
    from deeprl_signal_control_amd.agents import MA2C, VecA2C
    from deeprl_signal_control_amd.env import TrafficEnv, VecTrafficEnv, scenario_from_config
    from deeprl_signal_control_amd.trainer import VecTrainer
    cfg = _config()
    total_step = int(cfg.getfloat('TRAIN_CONFIG', 'total_step'))
    seed = cfg.getint('ENV_CONFIG', 'seed')
    # --- the reference's wiring (main.py:93-127) on the drop-in classes
    env = TrafficEnv(cfg['ENV_CONFIG'])
    assert env.T == 120 and env.test_num == 2 and sorted(env.nodes) == sorted(env.node_names)
    assert env.nodes['nt1'].ilds_in[0].endswith('_0') and len(env.nodes['nt1'].lanes_in) == 12
    model = MA2C(env.n_s_ls, env.n_a_ls, env.n_w_ls, env.n_f_ls, total_step, cfg['MODEL_CONFIG'], seed=seed)
    assert model.n_step == 40 and model.vec.cfg['lr_decay'] == 'linear' and model.vec.cfg['lr_min'] == 1e-5
    log = _reference_run(env, model, 2)
    # --- the batched trainer at E = 1, same seeds
    scn, s0, tseeds = scenario_from_config(cfg['ENV_CONFIG'])
    venv = VecTrafficEnv(scn, 1, seed=s0, test_seeds=tseeds, seed_stride=1)
    vmod = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, 1, scn.s_max, 5, cfg['MODEL_CONFIG'], total_step,
                  seed=seed, name='ma2c')
    tr = VecTrainer(venv, vmod)
    for _ in range(2 * 3):
        tr.run_iteration()
    torch.cuda.synchronize()
    assert model.vec.sample_step == vmod.sample_step == 240
    assert model.vec.lr_scheduler.n == vmod.lr_scheduler.n == 240
    # same actions -> same trajectories; the adaptor re-evaluates the forward graph for the update (unfused
    # kernels), the trainer uses the activations the fused forward cached: equal up to fp32 summation order
    pa, pb = model.vec.get_flat(), vmod.get_flat()
    assert np.abs(pa - pb).max() < 5e-5, np.abs(pa - pb).max()
    per_step = venv.reward_sum() / 240
    assert abs(per_step - np.mean([m for m, _ in log])) < 1e-9 * max(1.0, abs(per_step))
    env.close(); venv.close(); vmod.close(); model.vec.close()


def test_ia2c_adapter_and_make_env_and_test_seeds():
    from deeprl_signal_control_amd.agents import IA2C
    from deeprl_signal_control_amd.env import make_env
    cfg = _config()
    cfg['ENV_CONFIG']['agent'] = 'ia2c'
    cfg['MODEL_CONFIG']['reward_norm'] = '3000.0'
    env = make_env(cfg['ENV_CONFIG'])
    env.init_test_seeds([7, 8, 9])
    assert env.test_num == 3
    model = IA2C(env.n_s_ls, env.n_a_ls, env.n_w_ls, 1000, cfg['MODEL_CONFIG'], seed=3)
    env.train_mode = False
    ob = env.reset(test_ind=2)
    assert [len(o) for o in ob] == list(env.n_s_ls)
    model.reset()
    pol = model.forward(ob, True, 'p')
    assert all(abs(p.sum() - 1) < 1e-5 and len(p) == 5 for p in pol)
    ob2, r, done, g = env.step([int(np.argmax(p)) for p in pol])
    assert r.shape == (25,) and not done and isinstance(g, float)      # test mode: local rewards (envs/env.py:590-592)
    v = model.forward(ob2, False, 'v')
    assert len(v) == 25
    env.close(); model.vec.close()


def test_greedy_controllers_on_device(golden_dir):
    """tsc_env_greedy_actions (greedy_kernel over Scenario.greedy_controller_tables; VecTrafficEnv.greedy_actions) against the
    reference's three controllers:
    * LargeGridController (envs/large_grid_env.py:56-60, hard-coded lane sums) under the configs' norms (wave / 5 clipped at
      2): the controller sees float64 state, the kernel the float32 observation -- random counts incl. the float64
      non-tie 0.2 + 0.4 > 0.6 + 0 that IS a tie on the float32 values; then along a greedy-driven episode;
    * RealNetController: the fixture recorded from the reference class (tools/make_golden.py:greedy_fixtures; its waves are
      multiples of 0.1 = counts over norm_wave 10);
    * SmallGridController (STATE_PHASE_MAP) against the host restatement on float64 counts / 5."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net, build_small_grid
    from deeprl_signal_control_amd.trainer import greedy_actions
    from oracle.env_oracle import greedy_large_grid
    rng = np.random.RandomState(5)

    def on_device(env, ob64):
        ob = np.zeros((env.E, env.A, env.SMAX), np.float32)
        ob[:, :, :ob64.shape[2]] = ob64.astype(np.float32)
        return env.greedy_actions(torch.from_numpy(ob).cuda()).cpu().numpy()

    # large_grid, default norms
    scn = build_large_grid('greedy')
    assert (scn.norm_wave, scn.clip_wave) == (5.0, 2.0)
    E = 64
    env = VecTrafficEnv(scn, E, seed=40)
    cnt = rng.randint(0, 14, (E, 25, 6))
    cnt[0, 0] = [1, 0, 3, 2, 0, 0]             # phase 0: 0.2 + 0.4 (= 0.6000000000000001 in float64), phase 1: 0.6 + 0
    cnt[0, 1] = [3, 0, 1, 0, 0, 2]             # the other way round: phase 0 0.6 + 0, phase 1 0.2 + 0.4
    ob64 = np.clip(cnt / scn.norm_wave, 0, scn.clip_wave)
    want = np.array([[greedy_large_grid(ob64[e, a]) for a in range(25)] for e in range(E)])
    assert want[0, 0] == 0 and want[0, 1] == 1
    np.testing.assert_array_equal(on_device(env, ob64), want)
    np.testing.assert_array_equal(greedy_actions(scn, ob64), want)
    env.close()
    # ... and along an episode the controller itself drives (unit norms: the observations are the counts)
    scn = build_large_grid('greedy', norm_wave=1.0, norm_wait=1.0, clip_wave=-1.0, clip_wait=-1.0)
    E = 6
    env = VecTrafficEnv(scn, E, seed=40)
    env.train_mode = False
    ob = env.reset(test_ind=0)
    tot = 0.0
    for t in range(240):
        act = env.greedy_actions(ob)
        o = ob.cpu().numpy()
        want = np.array([[greedy_large_grid(o[e, a, :6]) for a in range(25)] for e in range(E)])
        np.testing.assert_array_equal(act.cpu().numpy(), want)
        ob, _, _, g = env.step(act)
        tot += float(g.mean().item())
    assert tot / 240 < -1.0
    env.close()
    # Monaco: the reference controller's own answers
    g = np.load(os.path.join(golden_dir, 'real_net_greedy_controller.npz'))
    scn = build_real_net('greedy', norm_wave=10.0, clip_wave=-1.0)
    env = VecTrafficEnv(scn, g['wave'].shape[0], seed=1)
    np.testing.assert_array_equal(on_device(env, g['wave']), g['action'])
    env.close()
    # small_grid
    scn = build_small_grid('greedy')
    E = 32
    env = VecTrafficEnv(scn, E, seed=1)
    n_own = max(len(v) for v in scn.extra['state_phase_map'].values())
    ob64 = np.clip(rng.randint(0, 12, (E, scn.n_agent, n_own)) / scn.norm_wave, 0, scn.clip_wave)
    np.testing.assert_array_equal(on_device(env, ob64), greedy_actions(scn, ob64))
    env.close()


def test_one_rank_rccl_group_reduces_the_library_buffer_in_place():
    """VecA2C.backward with torch.distributed initialised (1 rank, backend nccl = RCCL): the collective runs on the
    library's own gradient buffer (pointer equality) on the handle's stream, and the update equals the
    no-process-group update bit for bit."""
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.scenario import build_large_grid
    import tests.test_model_gpu as tm
    scn = build_large_grid('ma2c')
    E, T = 8, 5
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    res = []
    for with_pg in (False, True):
        if with_pg:
            torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        try:
            _, m, o = tm._make('ma2c', E, T, seed=4)
            gp, cnt = C.c_void_p(), C.c_int64()
            _lib.check(m._L.tsc_model_grad_buffer(m._h, C.byref(gp), C.byref(cnt)))
            assert m.grad_tensor().data_ptr() == gp.value and m.grad_tensor().numel() == cnt.value == m.n_param
            side = torch.cuda.Stream()
            m.use_stream(side)                                   # not torch's current stream
            m.reset(); o.reset()
            with torch.cuda.stream(side):
                obs, done = tm._fill(scn, m, o, E, T, np.random.RandomState(1), use_cache=True)
                R = m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone()
            m.backward(R)
            side.synchronize()
            res.append(m.get_flat().copy())
            m.close()
        finally:
            if with_pg:
                torch.distributed.destroy_process_group()
    np.testing.assert_array_equal(res[0], res[1])


def test_two_half_batch_handles_sum_to_the_full_batch_gradient():
    """The N-rank update rule on the HIP model: gradients of two handles fed the two halves of a batch, summed and
    scaled by 1/2, equal the gradient of one handle fed the whole batch (fp32 summation order apart)."""
    from deeprl_signal_control_amd import _lib
    import tests.test_model_gpu as tm
    E, T = 64, 6
    scn, full, _ = tm._make('ma2c', E, T, seed=8)
    _, ha, _ = tm._make('ma2c', E // 2, T, seed=8)
    _, hb, _ = tm._make('ma2c', E // 2, T, seed=8)
    np.testing.assert_array_equal(full.get_flat(), ha.get_flat())
    rng = np.random.RandomState(5)
    for m in (full, ha, hb):
        m.reset()
    done = np.ones(E, np.uint8)
    obs = tm._rand_obs(scn, E, rng)
    halves = ((ha, slice(0, E // 2)), (hb, slice(E // 2, E)))
    for t in range(T):
        act = rng.randint(0, 5, (E, 25)).astype(np.int32)
        rew = -rng.rand(E, 25) * 6000.0
        dpost = (rng.rand(E) < 0.15).astype(np.uint8)
        for m, sl in ((full, slice(0, E)),) + halves:
            o_, d_ = torch.from_numpy(obs[sl]).cuda(), torch.from_numpy(done[sl]).cuda()
            _, v, _ = m.forward_sample(o_, d_)
            m.add_transition(o_, d_, torch.from_numpy(act[sl]).cuda(), torch.from_numpy(rew[sl]).cuda(), v.clone(),
                             torch.from_numpy(dpost[sl]).cuda())
        obs, done = tm._rand_obs(scn, E, rng), dpost
    for m, sl in ((full, slice(0, E)),) + halves:
        R = m.forward(torch.from_numpy(obs[sl]).cuda(), False, 'v').clone()
        _lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(R.data_ptr()), 0.01))
    gf = full.grad_tensor().cpu().numpy()
    gs = 0.5 * (ha.grad_tensor() + hb.grad_tensor()).cpu().numpy()
    scale = np.abs(gf).max()
    assert scale > 0 and np.abs(gf - gs).max() <= 2e-5 * scale
    # ... and applying the summed buffer with scale 1/2 on both halves keeps the replicas identical
    g0 = ha.grad_tensor()
    g0.add_(hb.grad_tensor()); hb.grad_tensor().copy_(g0)
    ha._cur_lr = hb._cur_lr = 5e-4
    ha.apply_grads(0.5); hb.apply_grads(0.5)
    np.testing.assert_array_equal(ha.get_flat(), hb.get_flat())
    for m in (full, ha, hb):
        m.close()


@pytest.mark.gpu
def test_failing_creates_report_and_free_what_they_allocated():
    """An error return of tsc_env_create / tsc_model_create / tsc_iql_create hands out no handle, leaves the reason in
    tsc_last_error, and frees the handle with the device buffers it had already allocated (csrc/tsc_common.h CreateGuard):
    a scenario with a sibling lane out of range is refused AFTER the first lane tables were uploaded (four buffers); 2000 such
    calls -- 8000 buffers, at least 32 MB at the allocator's 4-KB granularity -- leave the device's free memory where it was."""
    import ctypes as C
    import torch
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.scenario import build_large_grid
    L = _lib.lib()
    scn = build_large_grid('ma2c')
    scn.lane_sib = scn.lane_sib.copy()
    scn.lane_sib[3] = scn.n_lane + 7
    sc, keep = _lib.scenario_struct(scn)
    torch.zeros(1).cuda()                                  # (the runtime's first allocation sets up its pools: ~150 MB once)
    assert L.tsc_env_create(C.byref(sc), 1024, 0, C.byref(C.c_void_p())) != 0
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(2000):
        h = C.c_void_p()
        assert L.tsc_env_create(C.byref(sc), 1024, 0, C.byref(h)) != 0 and not h.value
        assert b'sibling lane' in L.tsc_last_error()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (8 << 20)
    with pytest.raises(RuntimeError, match='sibling lane'):
        _lib.check(L.tsc_env_create(C.byref(sc), 4, 0, C.byref(C.c_void_p())))
