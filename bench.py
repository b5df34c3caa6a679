#!/usr/bin/env python
"""bench.py -- env-steps/s of the full on-policy A2C hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one A2C iteration of the reference's training loop (utils.py:284-295) over one batch
of env instances: n_step = 120 control steps (= 600 simulated seconds) of every env instance --
policy forward, fingerprint push, action sampling, microsimulator step, transition store -- plus
the n-step-return / BPTT / clip / RMSProp update (and, for N > 1, one RCCL all-reduce of the flat
gradient buffer).  Workload = BASELINE.json configs[2], the config its metric is quoted on:
large_grid 5x5, MA2C (FPLstmACPolicy), 1024 env instances per GPU, synthetic demand exactly as the
reference generator emits it, random-init (ortho) weights.

    value = agents x env instances (all ranks) x simulated seconds / wall time     [env-steps/s]

The timed region carries no profiling; a second pass of the same loop, HIP events around every launch on the launch
stream, gives the per-kernel table.  `--config c2 | c3 | c5` select BASELINE.json configs[1], [2] (default), [4].

Adds `roofline` (dominant kernel, timed live with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle -- C microsim + NumPy env wrapper + torch-CPU nets, float32 like the reference and float64
like the checker -- on a bounded sample of the same workload, rank 0, N = 1 only; it also quotes the committed timing of the
reference's own env class over the fake TraCI backend, which exists in the build container only).  The plain driver
line (`python bench.py`, N = 1) also carries short runs of the other single-GPU configurations, BASELINE.json configs[1]
(c2) and configs[4]'s per-GPU share (c5), under `extra.configs`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec


PROFILE_TAGS = ('r06', 'r05')           # committed rocprofv3 summaries this file may quote, newest first


def fetch_calibration():
    """(fetch factor, write factor, source) for 16-byte-per-lane streams from the committed calibration of the counters on a
    known byte count (tools/fetch_calib.hip -> profiles/rNN_fetch_calibration.json), or the guide's x2 on FETCH_SIZE and
    x1 on WRITE_SIZE when there is none."""
    for tag in PROFILE_TAGS:
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', '%s_fetch_calibration.json' % tag)))
            return (float(d['fetch']['read_f4']['factor']), float(d['write']['write_f4']['factor']),
                    'profiles/%s_fetch_calibration.json (1-GiB streams of 16 B per lane, tools/fetch_calib.hip)' % tag)
        except Exception:
            continue
    return 2.0, 1.0, 'MI355X_MICROARCH.md "HBM": FETCH_SIZE x 2 for 16 B/lane streams, WRITE_SIZE uncalibrated (x 1)'


def pmc_traffic(kernel, cfg_name, live):
    """(HBM bytes per launch of `kernel`, source) from the committed rocprofv3 --pmc summary of THIS configuration
    (profiles/rNN_pmc{,_c2,_c5}.json: separate FETCH_SIZE / WRITE_SIZE passes, KB units) or (None, reason).  The counters are
    CORRECTED by the factors the box reports on known 16-byte-per-lane streams (fetch_calibration: the simulator moves its
    vehicle state as 16-byte records since round 5, the update's kernels read 16 bytes per lane).  It is a COMMITTED
    measurement of an earlier run of this very command, not a counter read in this run; the counters average over whole
    episodes (tools/profile_round.sh) and the summary records the window-mean vehicles per instance of that run: a summary
    whose figure is more than 15 % away from this run's is REFUSED (the simulator's traffic scales with the vehicles)."""
    suffix = {'c3': '', 'c2': '_c2', 'c5': '_c5'}.get(cfg_name)
    if suffix is None:
        return None, 'no committed PMC summary for this configuration'
    why = 'no committed PMC summary (profiles/%s_pmc%s.json)' % (PROFILE_TAGS[0], suffix)
    ff, fw, fsrc = fetch_calibration()
    for tag in PROFILE_TAGS:
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', '%s_pmc%s.json' % (tag, suffix))))
            k = d['kernels'][kernel]
            v = float(d['mean_live_vehicles_per_env'])
        except Exception:
            continue
        if live and abs(v - live) > 0.15 * live:
            why = ('profiles/%s_pmc%s.json refused: collected at %.0f vehicles per instance, this run\'s window mean is %.0f '
                   '(more than 15 %% apart)' % (tag, suffix, v, live))
            continue
        return (ff * k['fetch_kb'] + fw * k['write_kb']) * 1024.0, (
            'profiles/%s_pmc%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over whole episodes of this '
            'configuration, window mean %.0f vehicles per instance against %.0f in this run): committed measurement, not '
            'collected in this run; FETCH_SIZE x %.3f, WRITE_SIZE x %.3f by %s' % (tag, suffix, v, live or 0.0, ff, fw, fsrc))
    return None, why


def rocprof_avg_us(kernel_key, cfg_name):
    """Average launch duration (us) of the kernel whose name contains `kernel_key` in the committed rocprofv3 --kernel-trace
    --stats summary of this configuration (profiles/rNN_kernel_stats{,_c2,_c5}.csv), or (None, None)."""
    import csv
    suffix = {'c3': '', 'c2': '_c2', 'c5': '_c5', 'q1': '_q1'}.get(cfg_name)
    if suffix is None:
        return None, None
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, 'profiles', '%s_kernel_stats%s.csv' % (tag, suffix))
        try:
            rows = [r for r in csv.DictReader(open(path)) if kernel_key in r['kernel']]
        except Exception:
            continue
        if rows:
            r = max(rows, key=lambda r_: float(r_['total_us']))
            return float(r['avg_us']), 'profiles/%s_kernel_stats%s.csv' % (tag, suffix)
    return None, None


def fused_update_kernels(model):
    """Which kernels the update of this model runs (mirrors tsc_model_create, csrc/tsc_model.hip): the one-pass kernels
    ('dwx_gemm' = dWx | dWh | dbl, 'dx1_gemm' = dX1 in registers + dW1 | db1 -- for the FcACPolicy also dWfc | dbfc) or the
    generic grouped GEMMs the profile ids are named after."""
    return model.H in (224, 160, 192, 128) and model.s_max <= 64 and model.s_max % 4 == 0


def algorithmic_flops(model, rows):
    """Per-launch ALGORITHMIC flops of every dense kernel for `rows` samples per agent-tower
    (per agent-tower MACs = n_wave*128 + n_fp*64 + n_wait*32 + H*256 + 64*256 + 64*n_out)."""
    fw, fp, ft = model.n_fc
    fc = sum(2 * (nw * fw + nf * fp + nt * ft) for nw, nf, nt in zip(model.n_wave_ls, model.n_f_ls, model.n_w_ls)) * 2
    H, L, G = model.H, model.Lh, model.G
    out = sum(2 * L * (na + 1) for na in model.n_a_ls)
    fused_update = fused_update_kernels(model)
    if model.policy == 'fc':            # FcACPolicy: the second layer is fc(H -> L), no recurrence (agents/policies.py:227-235)
        z = G * 2 * H * L
        if fused_update:                # fc_bwd_kernel ('dx1_gemm'): dX1 in registers -> dW1 | db1, and dWfc | dbfc, in one pass
            return {'policy_fwd_fused': (fc + z + out) * rows, 'fc_gemm': fc * rows, 'zx_gemm': z * rows, 'lstm_fwd': 0.0, 'lstm_bwd': 0.0,
                    'dwx_gemm': 0.0, 'dwh_gemm': 0.0, 'dx1_gemm': (2 * z + fc) * rows, 'dw1_gemm': 0.0, 'dwo_gemm': 0.0}
        return {'policy_fwd_fused': (fc + z + out) * rows, 'fc_gemm': fc * rows, 'zx_gemm': z * rows, 'lstm_fwd': 0.0, 'lstm_bwd': 0.0,
                'dwx_gemm': z * rows, 'dwh_gemm': 0.0, 'dx1_gemm': z * rows, 'dw1_gemm': fc * rows, 'dwo_gemm': 0.0}
    fused = fc + G * 2 * H * 4 * L + G * 2 * L * 4 * L + out           # one rollout forward of every tower
    # profile ids keep the names of the grouped GEMMs they started as: with the fused update kernels (default)
    # 'dwx_gemm' times dwxh_kernel (dWx + dWh + dbl in one pass) and 'dx1_gemm' times dx1w1_kernel2 (dX1 + dW1 + db1);
    # 'dwh_gemm' / 'dw1_gemm' are then only their split reductions
    dwx = G * 2 * H * 4 * L * rows
    dwh = G * 2 * L * 4 * L * rows
    return {'policy_fwd_fused': fused * rows, 'fc_gemm': fc * rows, 'zx_gemm': G * 2 * H * 4 * L * rows,
            'lstm_fwd': G * 2 * L * 4 * L * rows, 'lstm_bwd': G * 2 * L * 4 * L * rows,
            'dwx_gemm': dwx + dwh if fused_update else dwx, 'dwh_gemm': 0.0 if fused_update else dwh,
            'dx1_gemm': dwx + fc * rows if fused_update else dwx, 'dw1_gemm': 0.0 if fused_update else fc * rows,
            'dwo_gemm': 0.0}          # dWo = h^T dL is accumulated inside head_bwd (VALU); 'dwo_gemm' now times its slice reduction


def iql_algorithmic_flops(model, rows):
    """Per-launch ALGORITHMIC flops of the DeepQPolicy learner's kernels for `rows` rows per agent (agents/policies.py:343-371):
    one net evaluation = n_wave x 128 + n_wait x 32 + H1 x 64 + 64 x n_a MACs per row (W1's structural zeros excluded);
    'iql_grad' (one minibatch step) = Q(s') + Q(s) + the backward pass: dWq (64 x n_a), dX2 (64: dQ has ONE non-zero per row),
    dW2 and dX1 (H1 x 64 each), dW1 (n_wave x 128 + n_wait x 32); 'iql_act' = one evaluation of the acting rows."""
    lay = model.layout
    fwd = sum(nw * lay.n_fc0 + nt * lay.ft + lay.H1 * lay.H2 + lay.H2 * na for nw, nt, na in zip(model.n_wave_ls, model.n_w_ls, model.n_a_ls))
    bwd = sum(lay.H2 * na + lay.H2 + 2 * lay.H1 * lay.H2 + nw * lay.n_fc0 + nt * lay.ft for nw, nt, na in zip(model.n_wave_ls, model.n_w_ls, model.n_a_ls))
    return {'iql_grad': 2.0 * (2 * fwd + bwd) * rows, 'iql_act': 2.0 * fwd * rows}


def usable_cores():
    """Cores this process may actually run on: the affinity mask, cut by a cgroup CPU quota when one is visible."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: (lambda q, p_: None if q == 'max' else float(q) / float(p_))(*t.split()[:2])),
                        ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: None if int(t) <= 0 else int(t) / float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read()))):
        try:
            q = parse(open(path).read().strip())
            if q:
                cores = max(1, min(cores, int(q + 0.5)))
            break
        except Exception:
            continue
    return cores


def cpu_baseline(n_env=48, n_step=120, threads=None):
    """Same iteration on the host: oracle/ (test infrastructure) = C microsim + NumPy restatement of
    envs/env.py + torch-CPU restatement of agents/policies.py.  Bounded sample: n_env env instances,
    one iteration (n_step control steps + update), run twice: with float32 nets (the reference's TensorFlow arithmetic:
    `value`) and with the float64 nets the parity tests use as the checker (`value_float64_nets`).  SURVEY 8(d) baseline (i)
    -- the reference's OWN env class over the fake TraCI backend, which only exists in the build container -- is quoted
    from the committed profiles/r05_oracle_a.json (tools/time_oracle_a.py)."""
    from deeprl_signal_control_amd.scenario import build_large_grid
    from oracle.env_oracle import OracleEnv
    import oracle.nets_oracle as nets
    from oracle.nets_oracle import OracleA2C, choice_from_uniform
    from deeprl_signal_control_amd.agents import ortho_init
    # every core the process may use (the figure used 8 of the 16 the box offered), at most 32: the nets are
    # 64- to 224-wide matmuls on 48 x 25 rows, which stop scaling long before that
    torch.set_num_threads(max(1, min(threads or usable_cores(), 32)))
    scn = build_large_grid('ma2c')
    nw = [s - w - f for s, w, f in zip(scn.n_s_ls, scn.n_w_ls, scn.n_f_ls)]
    S = scn.s_max

    def towers():
        rng = np.random.RandomState(0)
        tw = []
        for a in range(scn.n_agent):
            for tower in ('pi', 'v'):
                p = {'fcw_w': ortho_init((nw[a], 128), rng), 'fcw_b': np.zeros(128, np.float32),
                     'fcf_w': ortho_init((scn.n_f_ls[a], 64), rng), 'fcf_b': np.zeros(64, np.float32),
                     'fct_w': ortho_init((scn.n_w_ls[a], 32), rng), 'fct_b': np.zeros(32, np.float32),
                     'lstm_wx': ortho_init((224, 256), rng), 'lstm_wh': ortho_init((64, 256), rng),
                     'lstm_b': np.zeros(256, np.float32)}
                n_out = scn.n_a_ls[a] if tower == 'pi' else 1
                p['out_w'] = ortho_init((64, n_out), rng); p['out_b'] = np.zeros(n_out, np.float32)
                tw.append(p)
        return tw

    def pack(obs):
        o = np.zeros((n_env, scn.n_agent, S))
        for e in range(n_env):
            for a in range(scn.n_agent):
                o[e, a, :len(obs[e][a])] = obs[e][a]
        return o

    def iteration(dtype):
        saved = nets.DT
        nets.DT = dtype                       # the restatement's working precision (module-level; restored below)
        try:
            rng = np.random.RandomState(0)
            envs = [OracleEnv(scn, seed=12 + e) for e in range(n_env)]
            model = OracleA2C(towers(), nw, scn.n_w_ls, scn.n_f_ls, scn.n_a_ls, n_env)
            t0 = time.perf_counter()
            obs = [e.reset() for e in envs]
            done = np.ones(n_env)
            for t in range(n_step):
                ob = pack(obs)
                pis, v = model.forward(ob, done, 'pv')
                acts = np.zeros((n_env, scn.n_agent), np.int64)
                rew = np.zeros((n_env, scn.n_agent))
                dpost = np.zeros(n_env)
                for e in range(n_env):
                    envs[e].update_fingerprint([pis[a][e] for a in range(scn.n_agent)])
                    acts[e] = [choice_from_uniform(pis[a][e], rng.rand()) for a in range(scn.n_agent)]
                    obs[e], r, d, _ = envs[e].step(list(acts[e]))
                    rew[e], dpost[e] = r, d
                model.add_transition(ob, done, acts, rew, v, dpost)
                done = dpost
            _, R = model.forward(pack(obs), np.zeros(n_env), 'v')
            grads, _ = model.compute_grads(R, 0.01)
            model.apply_grads(grads, 5e-4)
            return time.perf_counter() - t0
        finally:
            nets.DT = saved
    dt32 = iteration(torch.float32)
    dt64 = iteration(torch.float64)
    steps = scn.n_agent * n_env * n_step * scn.control_interval_sec
    # for scale: the C microsim alone (no env wrapper, no nets), one instance, one core, one full episode
    from oracle.microsim import MicroSim
    ms = MicroSim(scn)
    ms.reset(12)
    t1 = time.perf_counter()
    for t in range(0, 3600, 5):
        for a in range(scn.n_agent):
            ms.set_links(a, scn.phases[a][(t // 30) % 5])
        ms.step(5)
    sim_dt = time.perf_counter() - t1
    allc = sim_only_all_cores(scn)
    out = {'value': steps / dt32, 'unit': 'env-steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': 'oracle/ (C microsim + NumPy env wrapper + float32 torch-CPU nets = the reference\'s TensorFlow arithmetic): %d env '
                     'instances x %d control steps + 1 update, %.1f s' % (n_env, n_step, dt32),
           'value_float64_nets': steps / dt64,
           'sample_float64_nets': 'the same with the float64 nets the parity tests check against, %.1f s' % dt64,
           'sim_only_value': scn.n_agent * 3600 / sim_dt,
           'sim_only_sample': 'oracle/microsim.c alone, 1 instance, 1 core, 3600 simulated seconds, %.2f s' % sim_dt}
    if allc:
        out['sim_only_all_cores'] = allc
    try:                                # SURVEY 8(d) baseline (i), measured where /root/reference exists (not on this box)
        oa = json.load(open(os.path.join(ROOT, 'profiles', 'r05_oracle_a.json')))
        out['reference_env_over_fake_traci'] = {
            'source': 'profiles/r05_oracle_a.json (tools/time_oracle_a.py, build container, 1 core): committed measurement, NOT timed in this run',
            'env_only': {k: oa['oracle_a_env'][k] for k in ('value', 'unit', 'what')},
            'training_loop': {k: oa['oracle_a_training_loop'][k] for k in ('value', 'unit', 'what')}}
    except Exception:
        pass
    return out


def sim_only_all_cores(scn, target_s=3.0):
    """SURVEY 8(d) CPU baseline (ii): the C microsim alone over many env instances on ALL host cores -- one process per core
    (oracle/microsim_worker.py), every process a share of the instances, one full episode each under a fixed signal cycle,
    released together; value = agents x instances x simulated seconds / wall time from the release to the last answer."""
    import subprocess
    # a container usually sees every core of the host but may only run on a quota of them: one process per core it can USE
    cores = min(usable_cores(), 32)           # (a 256-thread GPU host ran 256 workers at the speed of ~9 cores: the quota is not always visible)
    per = max(1, int(target_s / 0.06))                     # one episode is ~0.04-0.07 s of one core
    procs = []
    try:
        for w in range(cores):
            procs.append(subprocess.Popen([sys.executable, '-m', 'oracle.microsim_worker', str(per), str(12 + w * per)], cwd=ROOT,
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for p in procs:
            if p.stdout.readline().strip() != 'ready':
                raise RuntimeError('worker did not come up')
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write('go\n'); p.stdin.flush()
        import select
        ans = []
        for p in procs:                                     # bounded: a box with fewer usable cores than it reports must not stall the line
            left = 60.0 - (time.perf_counter() - t0)
            if left <= 0 or not select.select([p.stdout], [], [], left)[0]:
                raise RuntimeError('workers did not finish within 60 s (%d processes x %d instances)' % (cores, per))
            ans.append(p.stdout.readline().split())
        dt = time.perf_counter() - t0
        live = float(np.mean([float(a[1]) for a in ans]))
        return {'value': scn.n_agent * cores * per * scn.episode_length_sec / dt, 'unit': 'env-steps/s', 'cores': cores,
                'sample': 'oracle/microsim.c alone (no env wrapper, no nets), %d processes x %d instances x one 3600-s episode under a '
                          'fixed 30-s signal cycle, %.1f s wall, slowest worker %.1f s, %.0f vehicles per instance on average'
                          % (cores, per, dt, max(float(a[0]) for a in ans), live)}
    except Exception as ex:                                 # a baseline leg must never take the bench line down
        return {'error': repr(ex)}
    finally:
        for p in procs:
            try:
                p.stdin.close(); p.wait(timeout=2)
            except Exception:
                p.kill()


def extra_lines(env, model, scn, n_ctrl=240):
    """SURVEY 8d asks for the simulator alone and simulator + policy forward next to the training figure: the same env
    instances, `n_ctrl` control steps each, actions pre-generated (uniform) for the sim-only line so that nothing but
    tsc_env_step is on the stream; env-steps/s = agents x instances x simulated seconds / wall time."""
    E, A, ctrl = env.E, scn.n_agent, scn.control_interval_sec
    out = {}
    # (a) the reference's greedy controller of the scenario as the action source, on the device (greedy_kernel): two launches
    # per control step, nothing pre-generated
    ob = env.reset(); model.reset()
    act = torch.zeros(E, A, dtype=torch.int32, device='cuda')
    for t in range(20):
        ob, _, _, _ = env.step(env.greedy_actions(ob, out=act))
    torch.cuda.synchronize()
    env.live_vehicle_mean(1)
    t0 = time.perf_counter()
    for t in range(n_ctrl):
        ob, _, _, _ = env.step(env.greedy_actions(ob, out=act))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['sim_only'] = {'value': A * E * n_ctrl * ctrl / dt, 'unit': 'env-steps/s', 'us_per_control_step': 1e6 * dt / n_ctrl,
                       'actions': 'the reference\'s greedy controller on the device (tsc_env_greedy_actions), one launch per control step',
                       'mean_live_vehicles_per_env': env.live_vehicle_mean(n_ctrl)}
    # (b) uniform random actions, pre-generated: nothing but tsc_env_step on the stream (more vehicles: random phases jam)
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    na = torch.as_tensor(np.asarray(scn.n_a_ls), device='cuda')
    acts = (torch.rand(16, E, A, generator=g, device='cuda') * na).to(torch.int32).contiguous()
    env.reset(); model.reset()
    for t in range(20):
        env.step(acts[t % 16])
    torch.cuda.synchronize()
    env.live_vehicle_mean(1)
    t0 = time.perf_counter()
    for t in range(n_ctrl):
        env.step(acts[t % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['sim_only_random_actions'] = {'value': A * E * n_ctrl * ctrl / dt, 'unit': 'env-steps/s', 'us_per_control_step': 1e6 * dt / n_ctrl,
                                      'actions': 'uniform random, pre-generated on the device',
                                      'mean_live_vehicles_per_env': env.live_vehicle_mean(n_ctrl)}
    ob = env.reset(); model.reset()
    done = True
    for t in range(20):
        pi, v, a = model.forward_sample(ob, done, cache=False)
        if scn.agent == 'ma2c':
            env.update_fingerprint(pi, zero_copy=True)
        ob, _, done, _ = env.step(a)
    torch.cuda.synchronize()
    env.live_vehicle_mean(1)
    t0 = time.perf_counter()
    for t in range(n_ctrl):
        pi, v, a = model.forward_sample(ob, done, cache=False)
        if scn.agent == 'ma2c':
            env.update_fingerprint(pi, zero_copy=True)
        ob, _, done, _ = env.step(a)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['sim_forward'] = {'value': A * E * n_ctrl * ctrl / dt, 'unit': 'env-steps/s', 'us_per_control_step': 1e6 * dt / n_ctrl,
                          'actions': 'sampled from the (random-init) policy: fused forward + sampling, fingerprint push, env step',
                          'mean_live_vehicles_per_env': env.live_vehicle_mean(n_ctrl)}
    return out


PRESETS = {'c2': ('large_grid', 'ia2c', 'fc', 256), 'c3': ('large_grid', 'ma2c', 'lstm', 1024), 'c5': ('real_net', 'ma2c', 'lstm', 512),
           'q1': ('large_grid', 'iqld', 'dqn', 1024)}      # q1: config/config_iqld_large.ini, not a BASELINE config


def iterations_per_episode(scenario, agent):
    """720 control steps per episode (3600 s / 5 s) over the agent's n_step (config/*.ini batch_size)."""
    n_step = 20 if agent in ('iqld', 'iqll') else 120 if scenario == 'large_grid' else 40
    return 720 // n_step


def warmup_iterations(ipe):
    """Whole warm-up episodes, at least 10 iterations (the first timed iterations of a fresh box otherwise run ~1 % below its
    steady clock): the timed window then starts at an episode boundary."""
    return ipe * max(1, -(-10 // ipe))


def preset_name(scenario, agent, policy, E):
    for k, v in PRESETS.items():
        if v == (scenario, agent, policy, E):
            return k
    return None


def rank_table(rank, world, local, backend):
    """What proves that N ranks ran on N devices: every rank's (rank, device index, device name, uuid / PCI bus id),
    all-gathered; rank 0 reports the list, the number of distinct devices, the backend and the RCCL version."""
    pr = torch.cuda.get_device_properties(local)
    ident = str(getattr(pr, 'uuid', '')) or str(getattr(pr, 'pci_bus_id', ''))
    try:
        ident += ' pci %04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    me = {'rank': rank, 'device_index': local, 'device_name': torch.cuda.get_device_name(local), 'device_id': ident,
          'host': os.uname().nodename, 'pid': os.getpid()}
    rows = [me]
    if world > 1:
        rows = [None] * world
        torch.distributed.all_gather_object(rows, me)
    try:
        ver = '.'.join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return {'world': world, 'backend': ('%s (RCCL %s)' % (backend, ver) if backend == 'nccl' else backend) if world > 1 else 'none (one rank)',
            'rccl_version': ver, 'distinct_devices': len({(r['host'], r['device_id'] or r['device_index']) for r in rows}), 'ranks': rows}


def run_config(args, rank, world, local, scenario, agent, policy, E, steps, warmup, want_extra, want_cpu, want_profile):
    """Build the env instances and the learner of one configuration on this rank's GPU, time `steps` A2C iterations after
    `warmup` (barrier + device synchronisation on both sides, MAX over ranks) and return the JSON line as a dict (rank 0)."""
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from deeprl_signal_control_amd.scenario import build_scenario
    from deeprl_signal_control_amd.trainer import MultiBatchTrainer, VecTrainer

    scn = build_scenario(scenario, agent, **({'lane_change': False} if (args.no_lane_change and scenario == 'large_grid') else {}))
    cfg_name = preset_name(scenario, agent, policy, E)
    is_q = agent in ('iqld', 'iqll')
    if is_q:                            # config/config_iql{d,l}_large.ini: batch 20, replay 1000, reward_norm 3000, Adam 1e-4
        mcfg, seed0, tseeds = dict(batch_size=20, buffer_size=1000, reward_norm=3000.0), 12, (10000, 20000)
    elif scenario == 'large_grid':      # config/config_{ma2c,ia2c}_large.ini
        mcfg, seed0, tseeds = dict(reward_norm=2000.0 if agent == 'ma2c' else 3000.0, batch_size=120), 12, (10000, 20000)
    else:                               # config/config_{ma2c,ia2c}_real.ini
        mcfg, seed0, tseeds = dict(reward_norm=1.0, batch_size=40), 42, (10000, 20000, 30000)
    B = max(1, args.batches)
    assert E % B == 0 and not (is_q and B > 1)
    Eb = E // B
    envs, models = [], []
    for b in range(B):
        envs.append(VecTrafficEnv(scn, Eb, device=local, seed=seed0 + rank * E + b * Eb, seed_stride=E * world,
                                  test_seeds=tseeds, resident=E))      # half-batches share the device: its load is E, not E / B
        # same weight-init seed on every rank / half-batch (replicas of one learner), own action stream each
        if is_q:
            from deeprl_signal_control_amd.iql import VecIQL
            mdl = VecIQL(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, Eb, scn.s_max, int(scn.green_tab.shape[1]), mcfg,
                         total_step=10 ** 6, device=local, seed=0, model_type='dqn' if agent == 'iqld' else 'lr')
        else:
            mdl = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, Eb, scn.s_max, int(scn.green_tab.shape[1]), mcfg,
                         device=local, seed=0, name=agent, policy=policy, replica=b)
        models.append(mdl)
    env, model = envs[0], models[0]
    tr = VecTrainer(env, model) if B == 1 else MultiBatchTrainer(envs, models)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    for _ in range(warmup):
        tr.run_iteration()
    sync()
    for e_ in envs:
        e_.live_vehicle_mean(1)                       # reset the window accumulators
    # ---- the timed region: profile-free (no event pairs between dependent launches) --------------------------------
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.run_iteration()
    sync()
    dt = time.perf_counter() - t0
    # window mean over the timed region of the vehicles in the network per env instance (SURVEY 8d's V)
    live = float(np.mean([e_.live_vehicle_mean(steps * model.n_step) for e_ in envs]))
    msr = tr.mean_step_reward()
    # ---- a second, profiled pass of the same loop: HIP events on the launch stream around every kernel -------------
    # a profiled pass covers one whole episode (T / n_step iterations, whatever its phase): its vehicle count is the episode mean,
    # the figure the committed PMC summaries were collected at
    prof, psteps, live_prof, dt_prof = {}, args.profile_steps or int(env.T // model.n_step), None, None
    if want_profile:
        # An event pair between two dependent launches costs the FOLLOWING launch (the packets behind the forward, which leaves
        # 160 MB of dirty lines, add ~17 us to the simulator step's figure; 98 against 80 us by rocprofv3), so the kernels
        # that alternate once per control step are bracketed in passes of their own: the simulator step alone, the other
        # per-control-step kernels (the rollout forward) alone, then the update's kernels together (tsc_profile_select).
        per_step = ('env_step', 'policy_fwd_fused', 'add_transition', 'fingerprint', 'sample', 'iql_act', 'iql_add')
        names = _lib.profile_names()
        passes = [['env_step'], [n for n in per_step if n != 'env_step'], [n for n in names if n not in per_step]]
        dt_prof, live_prof = 0.0, None
        for sel in passes:
            _lib.profile_select(sel)
            _lib.profile(enable=max(1, args.profile_stride), reset=True)
            t1 = time.perf_counter()
            for _ in range(psteps):
                tr.run_iteration()
            sync()
            dt_prof = max(dt_prof, time.perf_counter() - t1)
            got = _lib.profile()
            _lib.profile(enable=False)
            prof.update({k: v for k, v in got.items() if k in sel and v[0] > 0})
            lv = float(np.mean([e_.live_vehicle_mean(psteps * model.n_step) for e_ in envs]))
            if sel == ['env_step']:
                live_prof = lv                      # the vehicle count that belongs to the simulator's figure
        _lib.profile_select(None)
    extra = {}
    if rank == 0 and world == 1 and B == 1 and want_extra and not is_q:
        extra = extra_lines(env, model, scn)
    ranks = rank_table(rank, world, local, args.backend)          # collective: every rank
    if world > 1 and args.backend == 'nccl' and ranks['distinct_devices'] < world:
        # an RCCL line is a claim about N GPUs: ranks that share a device (a wrong LOCAL_RANK / visibility mask) void it
        raise SystemExit('bench.py: --backend nccl with %d ranks on %d distinct devices: %r' % (world, ranks['distinct_devices'], ranks['ranks']))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        n_step, ctrl = model.n_step, scn.control_interval_sec
        env_steps = scn.n_agent * E * world * n_step * ctrl * steps
        out = {'metric': 'env-steps/s (agents x envs x sim-steps/s), %s %s' % (scenario, agent.upper()),
               'value': env_steps / dt, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': steps,
               'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': ('%s (%d agents), %s (%s Q net), %d env instances per GPU; step = %d epsilon-greedy control steps '
                                       '(x%d sim-steps) of every instance into the replay rings + 10 minibatch steps (%d transitions per '
                                       'instance and agent each: TD loss, clip, Adam%s)'
                                       % ('large_grid 5x5', scn.n_agent, agent.upper(), policy.upper(), E, n_step, ctrl, n_step,
                                          (', RCCL grad all-reduce' if args.backend == 'nccl' else ', gloo grad all-reduce') if world > 1 else '')) if is_q else
                                      '%s (%d agents), %s %s policy%s, %d env instances per GPU; step = %d control steps '
                                      '(x%d sim-steps) of every instance + 1 A2C update (%sclip, RMSProp%s)'
                                      % ('large_grid 5x5' if scenario == 'large_grid' else 'real_net Monaco', scn.n_agent,
                                         agent.upper(), policy.upper(),
                                         ' (neighbour fingerprint gather)' if agent == 'ma2c' else '', E, n_step, ctrl,
                                         'BPTT, ' if policy == 'lstm' else '', (', RCCL grad all-reduce' if args.backend == 'nccl' else ', gloo grad all-reduce') if world > 1 else ''),
                          'envs_per_gpu': E, 'n_step': n_step, 'agents': scn.n_agent,
                          'parallelism': 'env-sharded x%d%s' % (world, ', %d half-batches on separate streams' % B if B > 1 else ''),
                          'mean_live_vehicles_per_env': live, 'live_vehicles': 'window mean over the timed region',
                          'mean_step_reward': msr}}
        if prof and is_q:
            # the Q learner: one fused kernel per minibatch step (csrc/tsc_iql_fused.h) dominates; its roofline is the f32 MFMA peak
            total = sum(ms for ms, _ in prof.values())
            kern = {k: {'ms_total': round(v[0], 3), 'launches': v[1]} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            out['kernels'] = kern
            if getattr(model, 'fused', False) and 'iql_grad' in prof:
                fl = iql_algorithmic_flops(model, E * n_step)['iql_grad']
                for k, rows_ in (('iql_grad', E * n_step), ('iql_act', E)):
                    if k in kern:
                        avg = kern[k]['ms_total'] / kern[k]['launches'] * 1e-3
                        kern[k]['frac_mfma'] = round(iql_algorithmic_flops(model, rows_)[k] / avg / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                ms, cnt = prof['iql_grad']
                ach = fl / (ms / cnt * 1e-3) / 1e12
                roof = {'bound': 'mfma', 'kernel': 'iql_grad', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                        'traffic_source': 'not collected: the kernel reads two 144-B observation rows per transition and writes its partial gradient once',
                        'algorithmic': '%.1f kFLOP per row and agent on average (two evaluations of the Q net + its backward pass, structural zeros '
                                       'of the block-diagonal first layer excluded; bench.py:iql_algorithmic_flops) x %d rows x %d agents'
                                       % (fl / (E * n_step) / scn.n_agent / 1e3, E * n_step, scn.n_agent),
                        'avg_launch_ms': ms / cnt, 'share_of_kernel_time': ms / total, 'kernel_time_ms_total': total,
                        'timed': 'HIP events on the launch stream around every launch of one group of kernels at a time, in passes of %d '
                                 'iterations right after the timed region (which carries no events)' % psteps}
                us, src = rocprof_avg_us('iql_fused_grad_kernel', cfg_name)
                if us:
                    roof['rocprofv3_avg_launch_ms'] = us * 1e-3
                    roof['rocprofv3_source'] = src + ' (committed rocprofv3 --kernel-trace --stats run of this command, not this run)'
                if 'env_step' in kern:
                    avg = kern['env_step']['ms_total'] / kern['env_step']['launches'] * 1e-3
                    alg = (32.0 * live_prof + 16.0 * scn.n_lane + scn.n_agent * 52.0 / 5.0) * ctrl * E
                    kern['env_step']['frac_hbm'] = round(alg / avg / 1e9 / PEAK_HBM_GBS, 4)
                out['roofline'] = roof
        elif prof:
            total = sum(ms for ms, _ in prof.values())
            # which kernel dominates: by event totals -- except that the simulator step's event figure carries the launch boundary
            # behind the forward (~3 us per launch when bracketed alone, ~17 us when every kernel is bracketed; the profiler's figure has
            # neither), which at 90 - 100 us per launch of either kernel can decide the order.  For the RANKING its per-launch time is capped at 1.1 x the committed
            # rocprofv3 average of this configuration's step kernel, when there is one; every reported figure stays the event's.
            rank = {k: v[0] for k, v in prof.items()}
            us_rp, _ = rocprof_avg_us('step_kernel', cfg_name)
            cap = {'applied': False}
            if us_rp and 'env_step' in rank:
                live_us = 1e3 * prof['env_step'][0] / prof['env_step'][1]
                if live_us > 1.35 * us_rp:
                    # the boundary explains ~3 - 17 us, not this: a live figure far above the committed one is a regression (or a stale
                    # profile) and must be able to make the simulator step the dominant kernel
                    cap['refused'] = 'live %.1f us per launch is more than 1.35 x the committed %.1f us' % (live_us, us_rp)
                elif 1.1 * us_rp < live_us:
                    uncapped_dom = max(rank, key=lambda k: rank[k])
                    rank['env_step'] = 1.1 * us_rp * 1e-3 * prof['env_step'][1]
                    cap = {'applied': True, 'env_step_us_live': live_us, 'env_step_us_ranked': 1.1 * us_rp,
                           'dominant_uncapped': uncapped_dom, 'changed_ranking': uncapped_dom != max(rank, key=lambda k: rank[k])}
            dom = max(rank, key=lambda k: rank[k])
            ms, cnt = prof[dom]
            avg_s = ms / cnt * 1e-3
            kern = {k: {'ms_total': round(v[0], 3), 'launches': v[1]} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            # per-kernel roofline fraction: MFMA kernels against the fp32 MFMA peak, the simulator against HBM
            fl_all = algorithmic_flops(model, E * n_step)
            for k, d in kern.items():
                avg = d['ms_total'] / d['launches'] * 1e-3
                if k == 'env_step':
                    alg = (32.0 * live_prof + 16.0 * scn.n_lane + scn.n_agent * 52.0 / 5.0) * ctrl * E      # SURVEY 8d bytes per launch
                    d['frac_hbm'] = round(alg / avg / 1e9 / PEAK_HBM_GBS, 4)
                    d['frac_hbm_timing'] = ('HIP-event pair around the launch, only this kernel bracketed in its pass: still includes the launch boundary '
                                            'behind the forward (~3 us against the whole-episode rocprofv3 average), i.e. understates the kernel')
                    us, src = rocprof_avg_us('step_kernel', cfg_name)
                    if us:      # the profiler's own figure of the same kernel in the committed trace of this configuration
                        d['rocprofv3_avg_us'] = us
                        d['frac_hbm_rocprofv3'] = round(alg / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
                        d['rocprofv3_source'] = src + ' (committed rocprofv3 --kernel-trace --stats run of this command, not this run)'
                    tb, tsrc = pmc_traffic('env_step', cfg_name, live_prof)
                    d['traffic'], d['traffic_source'] = tb, tsrc
                    if tb:
                        d['traffic_over_algorithmic'] = round(tb / alg, 3)
                elif k == 'policy_fwd_fused':
                    d['frac_mfma'] = round(algorithmic_flops(model, E)[k] / avg / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                elif fl_all.get(k, 0.0) > 0 and d['launches'] == psteps:
                    d['frac_mfma'] = round(fl_all[k] / avg / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            traffic, traffic_src = pmc_traffic(dom, cfg_name, live_prof)
            if dom == 'env_step':
                V, Ln, A = live_prof, scn.n_lane, scn.n_agent           # 32 V + 16 L + A*52/5 B per env-sim-step (V = window mean)
                bytes_launch = (32.0 * V + 16.0 * Ln + A * 52.0 / 5.0) * ctrl * E
                ach = bytes_launch / avg_s / 1e9
                roof = {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': ach / PEAK_HBM_GBS, 'traffic': traffic}
            else:
                fl = algorithmic_flops(model, E * n_step)
                if dom == 'policy_fwd_fused':                         # one launch per control step, rows = E
                    ach = algorithmic_flops(model, E)[dom] / avg_s / 1e12
                elif dom in ('fc_gemm', 'zx_gemm', 'lstm_fwd') and cnt != psteps:
                    # launched both per control step (rows = E) and once per update (rows = E * n_step)
                    tot_fl = fl[dom] * psteps + algorithmic_flops(model, E)[dom] * (cnt - psteps)
                    ach = tot_fl / (ms * 1e-3) / 1e12
                else:
                    ach = fl.get(dom, 0.0) / avg_s / 1e12
                roof = {'bound': 'mfma', 'kernel': dom, 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic}
            roof['traffic_source'] = traffic_src
            us, src = rocprof_avg_us({'env_step': 'step_kernel', 'policy_fwd_fused': 'policy_fwd_', 'dwx_gemm': 'dwxh_kernel',
                                      'dx1_gemm': 'fc_bwd_kernel' if policy == 'fc' else 'dx1w1_kernel2'}.get(dom, dom), cfg_name)
            if us:
                roof['rocprofv3_avg_launch_ms'] = us * 1e-3
                roof['rocprofv3_source'] = src + ' (committed rocprofv3 --kernel-trace --stats run of this command, not this run)'
            roof['timed'] = ('HIP events on the launch stream around every %slaunch of one group of kernels at a time (the simulator step / the '
                             'rollout forward / the update), in passes of %d iterations of the same loop right after the timed region (which '
                             'carries no events); the slowest pass took %.2f ms per iteration' % ('%d-th per-control-step ' % args.profile_stride if args.profile_stride > 1 else '',
                                                psteps, 1e3 * dt_prof / psteps))
            roof['mean_live_vehicles_per_env'] = live_prof
            roof['dominant_by'] = ('largest total of the HIP-event passes; the simulator step enters the ranking with at most 1.1 x its committed '
                                   'rocprofv3 average per launch (its event figure includes the launch boundary behind the forward)')
            roof['dominant_cap'] = cap
            roof['avg_launch_ms'] = ms / cnt
            roof['share_of_kernel_time'] = ms / total
            roof['kernel_time_ms_total'] = total
            out['roofline'] = roof
            out['kernels'] = kern
        extra['ranks'] = ranks
        out['extra'] = extra
        if world == 1 and want_cpu:
            out['cpu_baseline'] = cpu_baseline()
    for m_ in models:
        m_.close()
    for e_ in envs:
        e_.close()
    del tr, models, envs, env, model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed iterations (default: two episodes = 12 on large_grid, 36 on Monaco, 72 for IQL)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed iterations before (default: one episode, at least 10: a cold device needs ~0.3 s of load to reach its clocks)')
    ap.add_argument('--envs', type=int, default=1024, help='env instances per GPU')
    ap.add_argument('--agent', default='ma2c', choices=['ma2c', 'ia2c', 'iqld', 'iqll'])
    ap.add_argument('--scenario', default='large_grid', choices=['large_grid', 'real_net'])
    ap.add_argument('--policy', default='lstm', choices=['lstm', 'fc', 'dqn', 'lr'], help='fc = FcACPolicy (BASELINE configs[1], ia2c only); dqn / lr: the IQL agents\' Q nets')
    ap.add_argument('--batches', type=int, default=1, help='independent half-batches per GPU on separate HIP streams')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the sim-only / sim+forward lines (SURVEY 8d) and the other single-GPU configs')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-lane-change', action='store_true', help='large_grid without MICROSIM_SPEC.md rule 10 (the rounds 1 - 4 spec): A/B measurement only')
    ap.add_argument('--profile-stride', type=int, default=1,
                    help='profiled pass: HIP-event timing of every n-th launch of the per-control-step kernels (1 = all)')
    ap.add_argument('--profile-steps', type=int, default=0, help='iterations of every profiled pass (0 = one episode: T / n_step)')
    ap.add_argument('--config', default=None, choices=['c2', 'c3', 'c5', 'q1'],
                    help='BASELINE.json configs[i] presets: c2 = large_grid IA2C FC, 256 envs; c3 = large_grid MA2C LSTM, 1024 envs '
                         '(the default); c5 = real_net Monaco MA2C LSTM, 512 envs per GPU; q1 = large_grid IQL-DNN, 1024 envs (SURVEY 8f rank 1: 20 control '
                         'steps + 10 Adam minibatch steps per iteration)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend for N > 1 (nccl = RCCL; gloo only for the two-ranks-on-one-GPU test)')
    ap.add_argument('--device', type=int, default=None, help='force this device index on every rank (test only)')
    args = ap.parse_args()
    if args.config:
        args.scenario, args.agent, args.policy, args.envs = PRESETS[args.config]
    # the window covers >= 2 full episodes after >= 1 warm-up episode, so that the demand peak is inside and the window
    # mean of the vehicles is the episode mean (what the committed PMC passes were collected at).  An episode is 720 control steps =
    # 6 iterations of large_grid's A2C agents (n_step 120), 18 of Monaco's (40), 36 of the IQL agents (20).
    ipe = iterations_per_episode(args.scenario, args.agent)
    if args.steps is None:
        args.steps = 2 * ipe
    if args.warmup is None:
        args.warmup = warmup_iterations(ipe)
    # the plain driver line (no --config / --envs ... given) also carries the other single-GPU configurations of BASELINE.json
    plain = (args.config is None and (args.scenario, args.agent, args.policy, args.envs, args.batches) == ('large_grid', 'ma2c', 'lstm', 1024, 1))

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0)) if args.device is None else args.device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        if args.backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            torch.distributed.init_process_group('gloo')
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world)
    torch.cuda.set_device(local)

    out = run_config(args, rank, world, local, args.scenario, args.agent, args.policy, args.envs, args.steps, args.warmup,
                     want_extra=not args.no_extra, want_cpu=False, want_profile=not args.no_profile)
    if rank == 0 and world == 1 and plain and not args.no_extra:
        # BASELINE.json configs[1] and configs[4] (per-GPU share), same method, short runs: value, iteration time, window-mean
        # vehicles and the dominant kernel's roofline fraction of each -- one driver-run line evidences c2 / c3 / c5
        cfgs = {}
        for name in ('c2', 'c5'):
            sc, ag, po, E = PRESETS[name]
            o = run_config(args, rank, world, local, sc, ag, po, E, 2 * iterations_per_episode(sc, ag), warmup_iterations(iterations_per_episode(sc, ag)),
                           want_extra=False, want_cpu=False, want_profile=not args.no_profile)
            c = {'workload': o['config']['workload'], 'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'],
                 'steps': o['steps'], 'warmup': o['warmup'], 'mean_live_vehicles_per_env': o['config']['mean_live_vehicles_per_env']}
            if 'roofline' in o:
                r = o['roofline']
                c['roofline'] = {k: r[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'share_of_kernel_time')}
                c['kernels'] = o['kernels']
            cfgs[name] = c
        out.setdefault('extra', {})['configs'] = cfgs
        # the next row of SURVEY 8(f): the IQL-DNN learner on the same env path (config/config_iqld_large.ini), same method
        o = run_config(args, rank, world, local, *PRESETS['q1'], 2 * iterations_per_episode('large_grid', 'iqld'),
                       warmup_iterations(iterations_per_episode('large_grid', 'iqld')), want_extra=False, want_cpu=False,
                       want_profile=not args.no_profile)
        out['extra']['iql'] = {'workload': o['config']['workload'], 'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'],
                               'steps': o['steps'], 'warmup': o['warmup'], 'mean_live_vehicles_per_env': o['config']['mean_live_vehicles_per_env']}
        if 'roofline' in o:
            out['extra']['iql']['roofline'] = {k: o['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms',
                                                                              'share_of_kernel_time', 'algorithmic')}
            out['extra']['iql']['kernels'] = o['kernels']
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
