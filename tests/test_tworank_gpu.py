"""The N > 1 path EXECUTED on the HIP model: two processes, each with its own shard of env
instances, its own VecTrafficEnv + VecA2C + VecTrainer on the (one) GPU of the test box, exchanging nothing but the flat
gradient buffer.  The process group is gloo (two RCCL ranks cannot share one device; gloo moves device tensors for
broadcast / all_reduce): the product code path -- VecA2C.sync_replicas, VecA2C.backward -> allreduce_grads_ on the
library's own gradient buffer -> apply_grads(1 / world before the per-agent clip) -- is the one bench.py runs under RCCL.

Checked: sync_replicas makes rank 1 a replica of rank 0 although they were initialised from different seeds; parameters
and RMSProp accumulators stay BIT-identical across ranks over three iterations; the ranks explore independently (different
action streams); and the data-parallel update equals ONE process fed the same 2 x 16 instances as a single batch of 32
(same seeds, same actions) up to float32 summation order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

E, ITERS, SEED0 = 16, 3, 12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(rank, world, n_env, init_seed):
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from deeprl_signal_control_amd.scenario import build_scenario
    from deeprl_signal_control_amd.trainer import VecTrainer
    scn = build_scenario('large_grid', 'ma2c')
    env = VecTrafficEnv(scn, n_env, device=0, seed=SEED0 + rank * n_env, seed_stride=n_env * world)
    model = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, n_env, scn.s_max, int(scn.green_tab.shape[1]),
                   dict(batch_size=120), device=0, seed=init_seed, name='ma2c')
    return scn, env, model, VecTrainer(env, model)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    # different init seeds on purpose: sync_replicas (inside the constructor) must make rank 1 equal rank 0
    scn, env, model, tr = _build(rank, world, E, init_seed=1 + 7 * rank)
    assert model.world == world and model.rank == rank
    np.save(os.path.join(out, 'p_init_%d.npy' % rank), model.get_flat('params'))
    sl = model.rollout_slots()
    for it in range(ITERS):
        tr.start_episode() if tr.ob is None else None
        finished, R = tr.explore()
        if it == 0:
            np.save(os.path.join(out, 'act_%d.npy' % rank), sl['action'].cpu().numpy())
        model.compute_grads(R)
        local = model.grad_tensor().clone()
        from deeprl_signal_control_amd.agents import allreduce_grads_
        with torch.cuda.stream(model.stream):
            scale = allreduce_grads_(model.grad_tensor(), model.pg)        # what VecA2C.backward does
        assert scale == 1.0 / world
        if it == 0:
            np.save(os.path.join(out, 'glocal_%d.npy' % rank), local.cpu().numpy())
            np.save(os.path.join(out, 'gsum_%d.npy' % rank), model.grad_tensor().cpu().numpy())
        model.apply_grads(scale)
        np.save(os.path.join(out, 'p_%d_%d.npy' % (it, rank)), model.get_flat('params'))
        np.save(os.path.join(out, 'ms_%d_%d.npy' % (it, rank)), model.get_flat('ms'))
    # the packaged call (VecTrainer.run_iteration -> VecA2C.backward) once more, end to end
    tr.run_iteration()
    np.save(os.path.join(out, 'p_final_%d.npy' % rank), model.get_flat('params'))
    torch.distributed.destroy_process_group()


def test_two_ranks_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 29600 + os.getpid() % 300
    out = str(tmp_path)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    ld = lambda n: np.load(os.path.join(out, n))                       # noqa: E731
    np.testing.assert_array_equal(ld('p_init_0.npy'), ld('p_init_1.npy'))        # sync_replicas
    from deeprl_signal_control_amd.agents import init_tower_params  # noqa: F401
    for it in range(ITERS):
        np.testing.assert_array_equal(ld('p_%d_0.npy' % it), ld('p_%d_1.npy' % it))
        np.testing.assert_array_equal(ld('ms_%d_0.npy' % it), ld('ms_%d_1.npy' % it))
    np.testing.assert_array_equal(ld('p_final_0.npy'), ld('p_final_1.npy'))
    assert not np.array_equal(ld('p_final_0.npy'), ld('p_%d_0.npy' % (ITERS - 1)))
    np.testing.assert_array_equal(ld('gsum_0.npy'), ld('gsum_1.npy'))
    a0, a1 = ld('act_0.npy'), ld('act_1.npy')
    assert (a0 != a1).mean() > 0.3                                                # independent exploration
    # ---- one process, the same 32 instances as one batch, the recorded actions forced ------------------------------
    scn, env, model, tr = _build(0, 1, 2 * E, init_seed=1)
    np.testing.assert_array_equal(model.get_flat('params'), ld('p_init_0.npy'))
    sl = model.rollout_slots()
    acts = torch.from_numpy(np.concatenate([a0, a1], 1)).cuda()                   # [T, 2E, A]
    tr.start_episode()
    for t in range(model.n_step):
        pi, v, _ = model.forward_sample(sl['obs'][t], sl['done'][t], v_out=sl['value'][t], action_out=sl['action'][t])
        sl['action'][t].copy_(acts[t])
        env.update_fingerprint(pi, zero_copy=True)
        env.step(sl['action'][t], obs_out=sl['obs'][t + 1], reward_out=sl['reward'][t], done_out=sl['done'][t + 1])
        model.commit_transition()
    R = model.forward(sl['obs'][model.n_step], False, 'v')
    model.compute_grads(R)
    g = model.grad_tensor().cpu().numpy()
    g2 = 0.5 * ld('gsum_0.npy')
    G, stride = model.G, model.stride
    scale = np.abs(g.reshape(G, stride)).max(1, keepdims=True) + 1e-30
    assert (np.abs(g - g2).reshape(G, stride) / scale).max() < 2e-5
    np.testing.assert_allclose(0.5 * (ld('glocal_0.npy') + ld('glocal_1.npy')), g2, rtol=0, atol=1e-7 * np.abs(g2).max())
    model.apply_grads(1.0)
    np.testing.assert_allclose(model.get_flat('params'), ld('p_0_0.npy'), rtol=0, atol=2e-6)
    model.close(); env.close()


def test_bench_two_ranks_one_gpu():
    """bench.py's world > 1 branch (barrier, all-reduce inside backward, max-over-ranks timing) under the driver's own
    launch line, both ranks on device 0 over gloo (`--backend gloo --device 0`; the default is RCCL, one device per rank)."""
    port = 29950 + os.getpid() % 40
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
           '--envs', '64', '--backend', 'gloo', '--device', '0']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0 and d['steps'] == 1 and d['warmup'] == 1
    assert d['config']['envs_per_gpu'] == 64 and 'all-reduce' in d['config']['workload']
    assert d['config']['parallelism'].startswith('env-sharded x2')
    # whole-job figure: agents x instances of BOTH ranks x simulated seconds / the slower rank's time
    assert abs(d['value'] - 25 * 64 * 2 * 120 * 5 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert 'cpu_baseline' not in d and 'configs' not in d.get('extra', {})          # N = 1 only
    # the line proves how many ranks ran on how many devices (here: two ranks sharing device 0 over gloo)
    rk = d['extra']['ranks']
    assert rk['world'] == 2 and rk['backend'] == 'gloo' and rk['distinct_devices'] == 1
    assert sorted(r['rank'] for r in rk['ranks']) == [0, 1] and len({r['pid'] for r in rk['ranks']}) == 2
    assert all(r['device_index'] == 0 and r['device_name'] for r in rk['ranks'])
