"""N > 1 path on CPU (gloo, world_size 2): env sharding + the single gradient all-reduce
(DESIGN.md 6).  Each rank computes the oracle's gradients on ITS shard of env
instances; after `allreduce_grads_` + 1/world the flat buffer must equal the gradient of the global
batch, and both replicas must stay identical after the update."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from deeprl_signal_control_amd.agents import ParamLayout, allreduce_grads_, ortho_init, shard_seeds

A, T = 3, 4
NW, NT, NF, NA = [12, 18, 12], [6, 6, 6], [8, 12, 8], [5, 5, 5]
SMAX = 36


def _towers(rng):
    out = []
    for a in range(A):
        for tower in ('pi', 'v'):
            p = {'fcw_w': ortho_init((NW[a], 128), rng), 'fcw_b': np.zeros(128, np.float32),
                 'fcf_w': ortho_init((NF[a], 64), rng), 'fcf_b': np.zeros(64, np.float32),
                 'fct_w': ortho_init((NT[a], 32), rng), 'fct_b': np.zeros(32, np.float32),
                 'lstm_wx': ortho_init((224, 256), rng), 'lstm_wh': ortho_init((64, 256), rng),
                 'lstm_b': np.zeros(256, np.float32)}
            n_out = NA[a] if tower == 'pi' else 1
            p['out_w'] = ortho_init((64, n_out), rng); p['out_b'] = np.zeros(n_out, np.float32)
            out.append(p)
    return out


def _rollout(model, rng, E):
    done = np.ones(E)
    for t in range(T):
        obs = np.zeros((E, A, SMAX))
        for a in range(A):
            obs[:, a, :NW[a] + NT[a] + NF[a]] = rng.rand(E, NW[a] + NT[a] + NF[a])
        _, v = model.forward(obs, done, 'pv')
        act = rng.randint(0, 5, (E, A))
        rew = -rng.rand(E, A) * 4000
        dpost = (rng.rand(E) < 0.2).astype(np.float64)
        model.add_transition(obs, done, act, rew, v, dpost)
        done = dpost
    return np.zeros((E, A))


def _data(E_total):
    """Deterministic per-global-instance streams so shards see exactly their slice."""
    return [np.random.RandomState(1000 + e) for e in range(E_total)]


def _worker(rank, world, port, out):
    from oracle.nets_oracle import OracleA2C
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    lay = ParamLayout(NW, NT, NF, NA, SMAX, (128, 64, 32))
    model = OracleA2C(_towers(np.random.RandomState(0)), NW, NT, NF, NA, 1)
    R = _rollout(model, np.random.RandomState(1000 + rank), 1)          # this rank's env instance
    grads, _ = model.compute_grads(R, 0.01)
    flat = torch.from_numpy(lay.pack([{k: v.numpy() for k, v in g.items()} for g in grads]).copy())
    scale = allreduce_grads_(flat)
    assert scale == 1.0 / world
    g_avg = lay.unpack(flat.numpy() * scale)
    model.apply_grads([{k: torch.as_tensor(v, dtype=torch.float64) for k, v in g.items()} for g in g_avg], 5e-4)
    np.save(os.path.join(out, 'grad_%d.npy' % rank), flat.numpy() * scale)
    np.save(os.path.join(out, 'param_%d.npy' % rank), lay.pack(model.tower_params()))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    from oracle.nets_oracle import OracleA2C
    world, port = 2, 29517 + os.getpid() % 1000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (np.load(tmp_path / ('grad_%d.npy' % r)) for r in range(2))
    p0, p1 = (np.load(tmp_path / ('param_%d.npy' % r)) for r in range(2))
    np.testing.assert_array_equal(g0, g1)                  # same all-reduced buffer everywhere
    np.testing.assert_array_equal(p0, p1)                  # replicas stay identical
    # single process, global batch E = 2 (instance e uses the stream of rank e)
    lay = ParamLayout(NW, NT, NF, NA, SMAX, (128, 64, 32))
    model = OracleA2C(_towers(np.random.RandomState(0)), NW, NT, NF, NA, 2)

    class Both:                                             # interleave the two per-rank streams
        def __init__(self):
            self.r = [np.random.RandomState(1000), np.random.RandomState(1001)]

        def rand(self, E, n=None):
            return np.concatenate([r.rand(1) if n is None else r.rand(1, n) for r in self.r], 0)

        def randint(self, lo, hi, shape):
            return np.concatenate([r.randint(lo, hi, (1,) + tuple(shape[1:])) for r in self.r], 0)
    R = _rollout(model, Both(), 2)
    grads, _ = model.compute_grads(R, 0.01)
    ref = lay.pack([{k: v.numpy() for k, v in g.items()} for g in grads])
    np.testing.assert_allclose(g0, ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())   # float32 packing of the shards


def test_shard_seeds_are_a_partition():
    s = sum((shard_seeds(12, 4, r) for r in range(8)), [])
    assert s == list(range(12, 44))


def test_param_layout_roundtrip():
    lay = ParamLayout(NW, NT, NF, NA, SMAX, (128, 64, 32))
    tw = _towers(np.random.RandomState(3))
    flat = lay.pack(tw)
    back = lay.unpack(flat)
    for a, b in zip(tw, back):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k])
    assert flat.shape == (lay.n_param,)
    # structural zeros of the block-diagonal W1 are zero in the packed buffer
    W1 = flat.reshape(lay.G, lay.stride)[0, :lay.ob1].reshape(SMAX, lay.H)
    assert np.all(W1[:NW[0], 128:] == 0) and np.all(W1[NW[0]:NW[0] + NT[0], :192] == 0)


# ---- eight ranks (one node of MI355X as the driver launches it), gloo on CPU ------------------------------------------
def _worker8(rank, world, port, out):
    """What every rank of `bench.py --gpus 8` does around the device work: derive its env shard and its action stream, all-reduce
    the flat gradient buffer of the benchmarked model once per update, reduce the timing with MAX, resume from a checkpoint
    written by rank 0."""
    from deeprl_signal_control_amd.agents import (CKPT_FORMAT, replica_sample_seed, resume_sample_seed)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    dist = torch.distributed
    E, seed0, base = 1024, 12, 0
    seeds = shard_seeds(seed0, E, rank)
    sample_seed = replica_sample_seed(base, rank, 0)
    # the flat gradient buffer of large_grid MA2C (25 agents, both towers): one collective, in place
    lay = ParamLayout([12 + 6 * k for k in (2, 3, 3, 3, 2) * 5], [6] * 25, [8, 12, 12, 12, 8] * 5, [5] * 25, 52, (128, 64, 32))
    g = torch.full((lay.n_param,), float(rank + 1), dtype=torch.float32)
    g[rank::world] += 0.5                                    # rank-dependent pattern: a wrong reduction order / subset shows
    ptr = g.data_ptr()
    scale = allreduce_grads_(g)
    assert g.data_ptr() == ptr and scale == 1.0 / world
    # max-over-ranks timing (bench.py): every rank ends with the slowest rank's time
    t = torch.tensor([10.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # checkpoint written by rank 0 (format 2: the BASE seed), resumed by everyone
    path = os.path.join(out, 'checkpoint-120.npz')
    if rank == 0:
        np.savez(path, counters=np.array([77, base, 120, 120], np.int64), format=np.int64(CKPT_FORMAT))
        np.savez(os.path.join(out, 'old.npz'), counters=np.array([77, 424242, 120, 120], np.int64))
    dist.barrier()
    z, zo = np.load(path), np.load(os.path.join(out, 'old.npz'))
    b, resumed = resume_sample_seed(int(z['counters'][1]), z['format'], rank, 0)
    _, old = resume_sample_seed(int(zo['counters'][1]), None, rank, 0)
    np.save(os.path.join(out, 'r8_%d.npy' % rank),
            np.array([seeds[0], seeds[-1], sample_seed, resumed, b, old, int(t.item()), int(2 * float(g[0])), int(2 * float(g[1])), lay.n_param], np.int64))
    np.save(os.path.join(out, 'g8_%d.npy' % rank), g[:64].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_shards_streams_allreduce_and_resume(tmp_path):
    from deeprl_signal_control_amd.agents import replica_sample_seed
    world, port = 8, 30517 + os.getpid() % 1000
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / ('r8_%d.npy' % r)) for r in range(world)]
    # env shards: rank r owns global instances [r E, (r + 1) E) -> seeds seed0 + index, a partition of 8192 instances
    assert [(int(r[0]), int(r[1])) for r in rows] == [(12 + 1024 * k, 12 + 1024 * k + 1023) for k in range(world)]
    # action streams: rank 0 keeps the base seed (single-GPU runs are unchanged), all eight differ, a resume re-derives them
    streams = [int(r[2]) for r in rows]
    assert streams[0] == 0 and len(set(streams)) == world
    assert [int(r[3]) for r in rows] == [replica_sample_seed(0, k, 0) for k in range(world)] and all(int(r[4]) == 0 for r in rows)
    # a marker-less file (rounds 1 - 3): its slot holds the base seed too (rank 0 saved it), so every rank re-derives its own
    # stream from it -- eight distinct streams again, rank 0 on the stored value
    assert [int(r[5]) for r in rows] == [replica_sample_seed(424242, k, 0) for k in range(world)] and int(rows[0][5]) == 424242
    assert all(int(r[6]) == 10 + world - 1 for r in rows)      # MAX over ranks
    # the all-reduced buffer: sum_k (k + 1) = 36 everywhere, + 0.5 exactly once per element; identical on every rank
    g = [np.load(tmp_path / ('g8_%d.npy' % r)) for r in range(world)]
    for k in range(1, world):
        np.testing.assert_array_equal(g[0], g[k])
    np.testing.assert_array_equal(g[0], np.full(64, 36.5, np.float32))
    assert int(rows[0][9]) > 4_000_000                          # 17.3 MB padded
