"""Host-side env mirror of the reference's TrafficSimulator (envs/env.py:82-635) over the
HIP microsimulator (include/tsc.h, csrc/tsc_env.hip).

* ``VecTrafficEnv`` -- E parallel env instances on one GPU, torch tensors in/out
  (device memory only: no compute happens in torch).
* ``TrafficEnv`` -- E = 1 adaptor with the reference's exact duck-type
  (``reset/step/update_fingerprint/terminate`` and the attributes ``utils.Trainer`` and
  ``main.train`` read), so parity tests read like the reference's own
  scripts (envs/large_grid_env.py:261-286).

There is no CPU path: construction fails if libtsc.so is missing or no GPU is present.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .scenario import LANE_CAP, Scenario, build_scenario


class VecTrafficEnv:
    """E instances of TrafficSimulator.  Seeds follow the reference's bookkeeping
    (envs/env.py:547-560): instance e starts at ``seed + e`` and every ``reset()`` in train
    mode advances its seed by ``seed_stride`` (1 for E = 1, like the reference)."""

    def __init__(self, scn: Scenario, n_env: int, device=0, seed=12, test_seeds=(10000, 20000),
                 seed_stride=None, resident=None):
        if not torch.cuda.is_available():
            raise RuntimeError('VecTrafficEnv needs a GPU (MI355X); there is no CPU fallback')
        self.scn = scn
        self.E = int(n_env)
        self.device = torch.device('cuda', device) if not isinstance(device, torch.device) else device
        self.agent = scn.agent
        self.A, self.SMAX, self.AMAX = scn.n_agent, scn.s_max, int(scn.green_tab.shape[1])
        self.n_s_ls, self.n_a_ls = list(scn.n_s_ls), list(scn.n_a_ls)
        self.n_w_ls, self.n_f_ls = list(scn.n_w_ls), list(scn.n_f_ls)
        self.n_s, self.n_a = int(np.sum(self.n_s_ls)), int(np.prod(np.array(self.n_a_ls, dtype=object)))
        self.node_names = scn.node_names
        self.T = np.ceil(scn.episode_length_sec / scn.control_interval_sec)      # envs/env.py:89
        self.train_mode = True
        self.test_seeds = list(test_seeds)
        self.test_num = len(self.test_seeds)
        self.seeds = np.array([seed + e for e in range(self.E)], np.int64)
        self.seed_stride = self.E if seed_stride is None else seed_stride
        self.cur_episode = 0
        self.cur_sec = 0
        self._draws_routes = scn.stream_mode is not None and bool((np.asarray(scn.stream_mode) == 2).any())
        L = _lib.lib()
        self._L = L
        sc, self._keep = _lib.scenario_struct(scn)
        h = C.c_void_p()
        _lib.check(L.tsc_env_create(C.byref(sc), self.E, self.device.index or 0, C.byref(h)))
        self._h = h
        if resident is not None:        # env instances sharing the device with these (other handles / ranks), these included
            _lib.check(L.tsc_env_set_resident_instances(h, int(resident)))
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream(self.device)
            _lib.check(L.tsc_env_set_stream(h, C.c_void_p(self.stream.cuda_stream)))
            d = self.device
            # obs / done ping-pong: step() writes the buffers the previous call did NOT return, so the
            # caller can still hand the previous (ob, done) to model.add_transition (utils.py:160-165)
            self._obs2 = [torch.zeros(self.E, self.A, self.SMAX, dtype=torch.float32, device=d) for _ in range(2)]
            self._done2 = [torch.zeros(self.E, dtype=torch.uint8, device=d) for _ in range(2)]
            self._flip = 0
            self.obs, self.done = self._obs2[0], self._done2[0]
            self.reward = torch.zeros(self.E, self.A, dtype=torch.float64, device=d)
            self.global_reward = torch.zeros(self.E, dtype=torch.float64, device=d)

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None):
            self._L.tsc_env_destroy(self._h)
            self._h = None

    def terminate(self):
        """envs/env.py:563 closes the SUMO connection; the device-resident instances simply stay."""

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, stream):
        self.stream = stream
        _lib.check(self._L.tsc_env_set_stream(self._h, C.c_void_p(stream.cuda_stream)))

    def set_record(self, on, trip_cap=8192):
        """init_data(is_record=True) (envs/env.py:517-528): the following steps keep, per simulated second, the network
        statistics of _measure_traffic_step (:409-437), per control step the control log (:581-588) and a log of finished
        trips (:498-515), per env instance.  Uses the recording instantiation of the step kernel (plain lane walk)."""
        self.is_record = bool(on)
        self.trip_cap = int(trip_cap)
        if self.is_record or getattr(self, '_rec_alloc', False):
            _lib.check(self._L.tsc_env_record(self._h, int(self.is_record), int(trip_cap)))
            self._rec_alloc = True
        if self.is_record:
            K = self.scn.agent_lanes.shape[1]
            self._rec_ints = np.zeros((self.E, 8, 4), np.int64)
            self._rec_speed = np.zeros((self.E, 8), np.float64)
            self._rec_queue = np.zeros((self.E, 8, self.A * K), np.int32)
            self._ild_mask = (np.arange(K)[None, :] < np.asarray(self.scn.agent_nlane)[:, None]).ravel()
            self.traffic_data = [[] for _ in range(self.E)]
            self.control_data = [[] for _ in range(self.E)]
            self.trip_data = [[] for _ in range(self.E)]
            self.teleported_trips = [0] * self.E     # trips of the episode the teleport surrogate truncated (not in trip_data)
            self.truncated_trip_data = [[] for _ in range(self.E)]     # ... their rows (own table, see collect_tripinfo)

    def counters(self):
        """Per-instance counters of the running episode: (arrived, teleported), int64 [E] each.  `arrived` is the episode sum
        of simulation.getArrivedNumber (envs/env.py:413); `teleported` counts the heads the teleport surrogate removed after
        --time-to-teleport seconds (envs/env.py:283-284; SUMO would move them on, so they are not arrivals)."""
        arr, tel = np.zeros(self.E, np.uint64), np.zeros(self.E, np.uint64)
        _lib.check(self._L.tsc_env_counters(self._h, arr.ctypes.data_as(C.c_void_p), tel.ctypes.data_as(C.c_void_p)))
        return arr.astype(np.int64), tel.astype(np.int64)

    def _record_step(self, action):
        """Append this control step's rows (reference dict keys; envs/env.py:429-437, :582-587)."""
        vp = C.c_void_p
        _lib.check(self._L.tsc_env_read_record(self._h, self._rec_ints.ctypes.data_as(vp), self._rec_speed.ctypes.data_as(vp),
                                               self._rec_queue.ctypes.data_as(vp)))
        ctrl = self.scn.control_interval_sec
        act = action.cpu().numpy()
        g = self.global_reward.cpu().numpy()
        for e in range(self.E):
            for q in range(ctrl):
                n, dep, arr, wsum = (int(x) for x in self._rec_ints[e, q])
                queues = self._rec_queue[e, q][self._ild_mask]
                self.traffic_data[e].append({'episode': self.cur_episode, 'time_sec': self.cur_sec - ctrl + q + 1,
                                             'number_total_car': n, 'number_departed_car': dep, 'number_arrived_car': arr,
                                             'avg_wait_sec': wsum / n if n > 0 else 0,
                                             'avg_speed_mps': self._rec_speed[e, q] / n if n > 0 else 0,
                                             'std_queue': np.std(queues), 'avg_queue': np.mean(queues)})
            self.control_data[e].append({'episode': self.cur_episode, 'time_sec': self.cur_sec, 'step': self.cur_sec / ctrl,
                                         'action': ','.join(['%d' % a for a in act[e]]), 'reward': g[e]})

    def collect_tripinfo(self):
        """envs/env.py:498-515: the finished trips of the episode just run, in (arrival, route, serial) order; ids are
        f_<route>.<serial within the route>, times formatted like SUMO's tripinfo attributes."""
        buf = np.zeros((8192, 6), np.int32)
        for e in range(self.E):
            cnt = C.c_int32()
            _lib.check(self._L.tsc_env_read_trips(self._h, e, buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(cnt)))
            cap = getattr(self, 'trip_cap', len(buf))
            if cnt.value > cap:              # the kernel keeps the first trip_cap trips of an instance and counts the rest
                raise RuntimeError('instance %d finished %d trips, the trip log holds %d: call set_record(True, trip_cap=...) '
                                   'with a larger capacity' % (e, cnt.value, cap))
            if cnt.value > len(buf):
                buf = np.zeros((cnt.value, 6), np.int32)
                _lib.check(self._L.tsc_env_read_trips(self._h, e, buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(cnt)))
            tr = buf[:cnt.value]
            # a negative arrival marks a trip the teleport surrogate truncated (include/tsc.h tsc_env_read_trips): SUMO's
            # tripinfo file only lists vehicles that arrived, so the reference's collect_tripinfo never sees such a row
            self.teleported_trips[e] += int((tr[:, 3] < 0).sum())
            # ... but SUMO would have teleported the vehicle on and listed it later with a LONG duration: leaving these rows
            # out drops the worst-delayed trips of the episode, so they are kept apart (a fourth table, *_trip_truncated.csv,
            # with the time in the network and the waiting accumulated up to the removal) -- any trip-time / wait average
            # over the trip table alone is biased low by them
            cut = tr[tr[:, 3] < 0]
            cut = cut[np.lexsort((cut[:, 1], cut[:, 0], -cut[:, 3]))]
            for r, ser, dep, arr, wsec, wcnt in cut:
                self.truncated_trip_data[e].append({'episode': self.cur_episode, 'id': 'f_%d.%d' % (r, ser), 'depart_sec': '%.2f' % dep,
                                                    'removed_sec': '%.2f' % (-arr), 'duration_sec': '%.2f' % (-arr - dep),
                                                    'wait_step': '%d' % wcnt, 'wait_sec': '%.2f' % wsec})
            tr = tr[tr[:, 3] >= 0]
            tr = tr[np.lexsort((tr[:, 1], tr[:, 0], tr[:, 3]))]
            for r, ser, dep, arr, wsec, wcnt in tr:
                self.trip_data[e].append({'episode': self.cur_episode, 'id': 'f_%d.%d' % (r, ser), 'depart_sec': '%.2f' % dep,
                                          'arrival_sec': '%.2f' % arr, 'duration_sec': '%.2f' % (arr - dep),
                                          'wait_step': '%d' % wcnt, 'wait_sec': '%.2f' % wsec})

    def output_data(self, output_path, e=0, name=None):
        """envs/env.py:534-542: <name>_<agent>_{control,traffic,trip}.csv with the reference's columns (pandas'
        alphabetical order, as in real_net_experimental_data/eva_data/)."""
        import pandas as pd
        name = name or self.scn.name
        for kind, rows in (('control', self.control_data[e]), ('traffic', self.traffic_data[e]), ('trip', self.trip_data[e]),
                           ('trip_truncated', self.truncated_trip_data[e])):
            if kind == 'trip_truncated' and not rows:
                continue                                    # the reference's three tables always; the fourth only when it has rows
            df = pd.DataFrame(rows)
            df = df[sorted(df.columns)] if len(df.columns) else df
            df.to_csv(output_path + ('%s_%s_%s.csv' % (name, self.agent, kind)))

    def live_vehicle_mean(self, steps, reset=True):
        """Window mean of the vehicles in the network per env instance over the last `steps` control steps."""
        v = C.c_double()
        _lib.check(self._L.tsc_env_live_sum(self._h, C.byref(v), int(reset)))
        return v.value / max(1, steps * self.E)

    # -- reference API, batched ------------------------------------------------------------
    def reset(self, test_ind=0, obs_out=None):
        """envs/env.py:544-561 -> obs float32 [E, A, SMAX] (written to obs_out if given)."""
        self.obs = self._obs2[self._flip] if obs_out is None else obs_out
        if self.train_mode:
            seeds = self.seeds.copy()
            self.seeds += self.seed_stride                      # `self.seed += 1`, env.py:560
        else:
            # test_ind: one index for every instance (the reference's perform(test_ind)), or one index per
            # instance so that all test seeds are evaluated in one batched episode
            ti = np.broadcast_to(np.asarray(test_ind, np.int64), (self.E,))
            seeds = np.asarray(self.test_seeds, np.int64)[ti]
            self.seeds += self.seed_stride                      # the reference bumps it in test mode too
        s32 = (seeds & 0xFFFFFFFF).astype(np.uint32)
        if self._draws_routes:           # what gen_rou_file(seed) draws per episode (large_grid init_density > 0: the sinks)
            from .scenario import draw_stream_routes
            routes = np.ascontiguousarray(np.stack([draw_stream_routes(self.scn, int(sd)) for sd in seeds]), np.int32)
            _lib.check(self._L.tsc_env_set_stream_routes(self._h, routes.ctypes.data_as(C.POINTER(C.c_int32))))
        _lib.check(self._L.tsc_env_reset(self._h, s32.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         C.c_void_p(self.obs.data_ptr())))
        self.cur_sec = 0
        self.cur_episode += 1
        return self.obs

    def update_fingerprint(self, pi, zero_copy=False):
        """envs/env.py:633-635; pi float32 [E, A, AMAX] on the device.  zero_copy=True lets the env read
        `pi` in place at the next step() (the caller must not overwrite it before that)."""
        assert pi.dtype == torch.float32 and pi.is_contiguous() and tuple(pi.shape) == (self.E, self.A, self.AMAX)
        fn = self._L.tsc_env_bind_fingerprint if zero_copy else self._L.tsc_env_set_fingerprint
        _lib.check(fn(self._h, C.c_void_p(pi.data_ptr())))

    def greedy_actions(self, obs=None, out=None):
        """Controller.forward(obs) of the scenario's greedy controller (envs/large_grid_env.py:45-60,
        envs/real_net_env.py:78-111, envs/small_grid_env.py:40-55) for every instance, on the device (greedy_kernel,
        csrc/tsc_env.hip): obs float32 [E, A, SMAX] (default: the observation the env returned last) -> int32 [E, A]."""
        if not getattr(self, '_greedy_set', False):
            n_cand, term, action = self.scn.greedy_controller_tables()
            ip = C.POINTER(C.c_int32)
            n_cand, term, action = (np.ascontiguousarray(x, np.int32) for x in (n_cand, term, action))
            _lib.check(self._L.tsc_env_set_greedy(self._h, term.shape[1], term.shape[2], n_cand.ctypes.data_as(ip),
                                                  term.ctypes.data_as(ip), action.ctypes.data_as(ip)))
            self._greedy_set = True
        obs = self.obs if obs is None else obs
        assert obs.dtype == torch.float32 and obs.is_contiguous() and tuple(obs.shape) == (self.E, self.A, self.SMAX)
        if out is None:
            out = torch.empty(self.E, self.A, dtype=torch.int32, device=self.device)
        assert out.dtype == torch.int32 and out.is_contiguous() and tuple(out.shape) == (self.E, self.A)
        _lib.check(self._L.tsc_env_greedy_actions(self._h, C.c_void_p(obs.data_ptr()), C.c_void_p(out.data_ptr())))
        return out

    def reward_sum(self, reset=False):
        """Sum of the global reward over instances and control steps since the accumulator was reset."""
        v = C.c_double()
        _lib.check(self._L.tsc_env_reward_sum(self._h, C.byref(v), int(reset)))
        return v.value

    def step(self, action, obs_out=None, reward_out=None, done_out=None):
        """envs/env.py:566-631; action int32 [E, A] on the device.  Returns the env's own
        output buffers (obs f32 [E,A,SMAX], reward f64 [E,A], done u8 [E], global f64 [E]);
        obs / done alternate between two buffers, reward / global are overwritten by the next call.
        obs_out / reward_out / done_out: write there instead (e.g. the learner's rollout slots, zero copy)."""
        assert action.dtype == torch.int32 and action.is_contiguous() and tuple(action.shape) == (self.E, self.A)
        self._flip ^= 1
        self.obs = self._obs2[self._flip] if obs_out is None else obs_out
        self.done = self._done2[self._flip] if done_out is None else done_out
        reward = self.reward if reward_out is None else reward_out
        _lib.check(self._L.tsc_env_step(self._h, C.c_void_p(action.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                        C.c_void_p(reward.data_ptr()),
                                        C.c_void_p(self.global_reward.data_ptr()),
                                        C.c_void_p(self.done.data_ptr()), int(self.train_mode)))
        self.cur_sec += self.scn.control_interval_sec
        if getattr(self, 'is_record', False):
            self._record_step(action)
        return self.obs, reward, self.done, self.global_reward

    # -- debug / parity ---------------------------------------------------------------------
    def get_state(self, e=0):
        NL, NR = self.scn.n_lane, self.scn.n_stream          # pending / serial are per insertion stream
        out = dict(n=np.zeros(NL, np.int32), x=np.zeros((NL, LANE_CAP), np.float32),
                   v=np.zeros((NL, LANE_CAP), np.float32), sf=np.zeros((NL, LANE_CAP), np.float32),
                   w=np.zeros((NL, LANE_CAP), np.int32), r=np.zeros((NL, LANE_CAP), np.int32),
                   pending=np.zeros(NR, np.int32), serial=np.zeros(NR, np.int32), t=np.zeros(1, np.int32))
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        _lib.check(self._L.tsc_env_get_state(
            self._h, e, out['n'].ctypes.data_as(ip), out['x'].ctypes.data_as(fp), out['v'].ctypes.data_as(fp),
            out['sf'].ctypes.data_as(fp), out['w'].ctypes.data_as(ip), out['r'].ctypes.data_as(ip),
            out['pending'].ctypes.data_as(ip), out['serial'].ctypes.data_as(ip), out['t'].ctypes.data_as(ip)))
        return out

    def mean_live_vehicles(self):
        v = C.c_double()
        _lib.check(self._L.tsc_env_live_vehicles(self._h, C.byref(v)))
        return v.value


ENV_CONFIG_KEYS = dict(control_interval_sec=int, yellow_interval_sec=int, episode_length_sec=int, coop_gamma=float,
                       norm_wave=float, norm_wait=float, clip_wave=float, clip_wait=float, coef_wait=float,
                       objective=str)
SCENARIO_KEYS = {'large_grid': dict(peak_flow1=int, peak_flow2=int, init_density=float),
                 'real_net': dict(flow_rate=int), 'small_grid': dict(num_extra_car_per_hour=int)}


def scenario_from_config(config):
    """[ENV_CONFIG] section (configparser SectionProxy or dict of strings, config/config_*.ini) -> (Scenario, seed,
    test_seeds): the keys TrafficSimulator.__init__ and the scenario subclasses read (envs/env.py:83-110,
    envs/large_grid_env.py:64-68, envs/real_net_env.py:115-117, envs/small_grid_env.py:59-61).  `data_path` is
    accepted and ignored: nothing is generated on disk (the demand tables live on the device)."""
    get = config.get
    name, agent = get('scenario'), get('agent')
    kw = {}
    for k, typ in list(ENV_CONFIG_KEYS.items()) + list(SCENARIO_KEYS.get(name, {}).items()):
        v = get(k)
        if v is not None:
            kw[k] = typ(float(v)) if typ is int else typ(v)
    seed = int(get('seed'))
    test_seeds = tuple(int(x) for x in str(get('test_seeds')).split(','))
    return build_scenario(name, agent, **kw), seed, test_seeds


class NodeView:
    """Read-only stand-in for envs/env.py:63-80 `Node`: what main.init_env hands to the greedy controllers
    (envs/real_net_env.py:78-111 reads lanes_in / ilds_in) and what the evaluation scripts print."""

    def __init__(self, scn, a):
        self.name = scn.node_names[a]
        self.control = False
        self.lanes_in = [scn.lane_names[l] for l in scn.link_lane[a, :scn.agent_nlink[a]]]
        self.ilds_in = [scn.lane_names[l] for l in scn.agent_lanes[a, :scn.agent_nlane[a]]]
        self.neighbor = [scn.node_names[j] for j in scn.neighbors[a]]
        self.n_a = int(scn.n_a_ls[a])
        self.num_state = int(scn.agent_nlane[a])
        self.num_fingerprint = int(scn.n_f_ls[a])
        self.phase_id = scn.extra.get('phase_ids', {}).get(self.name, scn.name)
        self.phases = list(scn.phases[a])


class TrafficEnv:
    """The reference's single-env duck-type (envs/env.py:544-635) on top of VecTrafficEnv(E=1):
    ``reset() -> list[A] of 1-D float arrays``, ``step(list[A] of int) -> (obs, reward ndarray,
    done bool, global_reward float)``.  Observations come back as float32 (the value the
    reference feeds to its nets, agents/policies.py:82,130); rewards stay float64.

    Constructed like the reference's env classes -- ``TrafficEnv(config['ENV_CONFIG'], port=0, output_path='',
    is_record=False, record_stat=False)`` (envs/large_grid_env.py:64-68) -- or from a compiled Scenario."""

    def __init__(self, config, port=0, output_path='', is_record=False, record_stat=False, device=0, seed=None,
                 test_seeds=None):
        if isinstance(config, Scenario):
            scn, cseed, ctest = config, 12, (10000, 20000)
        else:
            scn, cseed, ctest = scenario_from_config(config)
        seed = cseed if seed is None else seed
        test_seeds = ctest if test_seeds is None else test_seeds
        self.vec = VecTrafficEnv(scn, 1, device=device, seed=seed + port, test_seeds=test_seeds, seed_stride=1)
        self.scn = scn
        for k in ('agent', 'n_s_ls', 'n_a_ls', 'n_w_ls', 'n_f_ls', 'n_s', 'n_a', 'node_names', 'T'):
            setattr(self, k, getattr(self.vec, k))
        self.name = scn.name
        self.port = port
        self.nodes = {n: NodeView(scn, a) for a, n in enumerate(scn.node_names)}
        self.init_data(is_record, record_stat, output_path)

    test_num = property(lambda self: self.vec.test_num)
    test_seeds = property(lambda self: self.vec.test_seeds)

    def init_test_seeds(self, test_seeds):
        """envs/env.py:530-532."""
        self.vec.test_seeds = [int(s) for s in test_seeds]
        self.vec.test_num = len(self.vec.test_seeds)

    def init_data(self, is_record, record_stats, output_path):
        """envs/env.py:517-528."""
        self.is_record, self.record_stats, self.output_path = is_record, record_stats, output_path
        self.vec.set_record(bool(is_record))

    traffic_data = property(lambda self: self.vec.traffic_data[0])
    control_data = property(lambda self: self.vec.control_data[0])
    trip_data = property(lambda self: self.vec.trip_data[0])
    truncated_trip_data = property(lambda self: self.vec.truncated_trip_data[0])

    def collect_tripinfo(self):
        """envs/env.py:498-515 (call after the episode, like the reference's evaluation scripts)."""
        self.vec.collect_tripinfo()

    def output_data(self):
        """envs/env.py:534-542."""
        if not self.is_record:
            raise RuntimeError('Env: no record to output!')
        self.vec.output_data(self.output_path, 0, self.name)

    train_mode = property(lambda self: self.vec.train_mode,
                          lambda self, v: setattr(self.vec, 'train_mode', v))
    cur_episode = property(lambda self: self.vec.cur_episode)
    cur_sec = property(lambda self: self.vec.cur_sec)
    seed = property(lambda self: int(self.vec.seeds[0]))

    def _split(self, obs):
        o = obs[0].cpu().numpy()
        return [o[a, :n].copy() for a, n in enumerate(self.scn.obs_len)]

    def reset(self, gui=False, test_ind=0):
        return self._split(self.vec.reset(test_ind=test_ind))

    def update_fingerprint(self, policy):
        pi = np.zeros((1, self.vec.A, self.vec.AMAX), np.float32)
        for a, p in enumerate(policy):
            pi[0, a, :len(p)] = np.asarray(p, np.float32)
        self.vec.update_fingerprint(torch.from_numpy(pi).to(self.vec.device))

    def step(self, action):
        act = torch.tensor([[int(a) for a in action]], dtype=torch.int32, device=self.vec.device)
        obs, reward, done, g = self.vec.step(act)
        r = reward[0].cpu().numpy()
        if self.agent in ('a2c', 'greedy') and self.train_mode:
            r = float(r[0])                                    # scalar for greedy/a2c (envs/env.py:593-594)
        return self._split(obs), r, bool(done[0].item()), float(g[0].item())

    def terminate(self):
        pass

    def close(self):
        self.vec.close()


def make_env(scenario='large_grid', agent='ma2c', n_env=None, **kw):
    """init_env (main.py:51-79) equivalent; n_env=None gives the reference's single-env type.  `scenario` may also be
    the [ENV_CONFIG] section itself (then `agent` and the other keys come from it)."""
    if not isinstance(scenario, str):
        scn, seed, test_seeds = scenario_from_config(scenario)
        kw.setdefault('seed', seed); kw.setdefault('test_seeds', test_seeds)
        return TrafficEnv(scn, **kw) if n_env is None else VecTrafficEnv(scn, n_env, **kw)
    scn_kw = {k: kw.pop(k) for k in list(kw)
              if k in Scenario.__dataclass_fields__ or k in ('peak_flow1', 'peak_flow2', 'sort_lanes')}
    scn = build_scenario(scenario, agent, **scn_kw)
    return TrafficEnv(scn, **kw) if n_env is None else VecTrafficEnv(scn, n_env, **kw)
