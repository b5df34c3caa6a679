"""NumPy restatement of the reference env wrapper (envs/env.py) over the CPU
microsim -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Unlike oracle/fake_traci.py (which runs the reference's own classes and only
exists in the build container), this file travels to the GPU box and is the
checker the `-m gpu` parity tests compare the HIP path against.  It is pinned
against the reference's own classes by tests/test_oracle_golden.py using the
fixtures tools/make_golden.py wrote under tests/golden/.

Every method cites the reference lines it follows.
"""
import numpy as np

from oracle.microsim import MicroSim

REALNET_REWARD_NORM = 20            # envs/env.py:18


def numpy_pairwise_sum(a):
    """The order np.sum uses for n <= 128 float64 (numpy loops_utils pairwise sum,
    8 accumulators); `global_reward = np.sum(reward)` at envs/env.py:580 depends on it.
    Restated here in plain Python so tests can pin the HIP kernel's order."""
    a = [float(x) for x in a]
    n = len(a)
    if n < 8:
        r = 0.0
        for x in a:
            r += x
        return r
    r = a[:8]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] += a[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res += a[i]
        i += 1
    return res


def episode_stream_routes(scn, seed):
    """What gen_rou_file(seed) draws per episode beyond the fixed flow table (large_grid/data/build_file.py:223-266,275-281:
    np.random.seed(seed), then ONE np.random.choice(sink_edges) per initial flow, in flow order): the route of every stream whose
    route is the generator's draw (stream mode 2), int32 [NS], or None.  The checker's own restatement -- one scalar draw per flow,
    like the reference -- so that a slip in the product's vectorised draw (scenario.draw_stream_routes) shows up as a HIP-vs-oracle
    difference; tests/test_init_density.py pins both against the reference generator's file."""
    if scn.stream_mode is None or not (np.asarray(scn.stream_mode) == 2).any():
        return None
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    routes = np.asarray(scn.stream_choice)[:, 0, 0, 0].astype(np.int32).copy()
    for s_ in range(len(routes)):
        if int(scn.stream_mode[s_]) != 2:
            continue
        cand = [int(r) for r in np.asarray(scn.stream_choice)[s_, 0, :, 0] if r >= 0]
        routes[s_] = cand[int(rs.choice(len(cand)))]
    return routes


class OracleEnv:
    """E = 1 restatement of TrafficSimulator (envs/env.py:82-635) for MARL agents."""

    def __init__(self, scn, seed=12, test_seeds=(10000, 20000), train_mode=True, is_record=False):
        self.scn = scn
        self.is_record = is_record                     # env.py:517-528
        self.traffic_data, self.control_data, self.trip_data = [], [], []
        self.agent = scn.agent
        self.seed = seed
        self.test_seeds = list(test_seeds)
        self.train_mode = train_mode
        self.ms = MicroSim(scn)
        if is_record:
            self.ms.record()
        self.n_agent = scn.n_agent
        self.node_names = scn.node_names
        self.n_s_ls, self.n_a_ls = scn.n_s_ls, scn.n_a_ls
        self.n_w_ls, self.n_f_ls = scn.n_w_ls, scn.n_f_ls
        self.T = np.ceil(scn.episode_length_sec / scn.control_interval_sec)   # env.py:89
        self.cur_episode = 0
        self.prev_action = [0] * self.n_agent
        self.fingerprint = [np.zeros(n - 1) for n in self.n_a_ls]

    # -- env.py:128-152 via the precompiled table (scenario.yellow_phase restates the rule)
    def _phase(self, a, action, phase_type):
        scn = self.scn
        k = scn.agent_nlink[a]
        if phase_type == 'green':
            return bytes(scn.green_tab[a, action, :k])
        prev = self.prev_action[a]
        self.prev_action[a] = action
        if prev < 0 or action == prev:
            return bytes(scn.green_tab[a, action, :k])
        return bytes(scn.yellow_tab[a, prev, action, :k])

    def _set_phase(self, action, phase_type):             # env.py:455-459
        for a, act in enumerate(action):
            self.ms.set_links(a, self._phase(a, int(act), phase_type))

    def _measure(self):
        """env.py:369-407 (state) and :325-367 (reward) share the detector read-out."""
        scn = self.scn
        wave, wait, halt = [], [], []
        for a in range(self.n_agent):
            w_, t_, h_ = [], [], []
            for k in range(scn.agent_nlane[a]):
                l = scn.agent_lanes[a, k]
                w, h, hw = self.ms.lane_stats(l)
                w_.append(w)
                t_.append(hw)
                h_.append(h)
            wave.append(np.array(w_))
            wait.append(np.array(t_))
            halt.append(np.array(h_))
        return wave, wait, halt

    @staticmethod
    def _norm_clip(x, norm, clip):                         # env.py:439-442
        x = x / norm
        return x if clip < 0 else np.clip(x, 0, clip)

    def _get_state(self, wave, wait):                      # env.py:163-205
        scn = self.scn
        wave_s = [self._norm_clip(w, scn.norm_wave, scn.clip_wave) for w in wave]
        wait_s = [self._norm_clip(w, scn.norm_wait, scn.clip_wait) for w in wait]
        state = []
        for a in range(self.n_agent):
            if self.agent == 'greedy':
                state.append(wave_s[a])
                continue
            cur = [wave_s[a]]
            for j in scn.neighbors[a]:
                cur.append(wave_s[j] * scn.coop_gamma if self.agent == 'ma2c' else wave_s[j])
            if scn.has_wait_state:
                cur.append(wait_s[a])
            if self.agent == 'ma2c':
                for j in scn.neighbors[a]:
                    cur.append(self.fingerprint[j])
            state.append(np.concatenate(cur))
        return state

    def _reward(self, wait, halt):                         # env.py:325-367
        scn = self.scn
        rewards = []
        for a in range(self.n_agent):
            q = halt[a]
            if scn.queue_cap >= 0:
                q = np.minimum(scn.queue_cap, q)
            queue = np.sum(q) if scn.objective in ('queue', 'hybrid') else 0
            w = np.sum(wait[a]) if scn.objective in ('wait', 'hybrid') else 0
            if scn.objective == 'queue':
                r = - queue
            elif scn.objective == 'wait':
                r = - w
            else:
                r = - queue - scn.coef_wait * w
            rewards.append(r)
        return np.array(rewards)

    def _simulate(self, num_step):                         # env.py:461-471
        for _ in range(num_step):
            self.ms.step(1)
            self.cur_sec += 1
            if self.is_record:
                self._measure_traffic_step()

    def _measure_traffic_step(self):                       # env.py:409-437
        scn = self.scn
        n, wsum, ssum = self.ms.network_stats()
        tot = self.ms.totals()
        queues = []
        for a in range(self.n_agent):
            for k in range(scn.agent_nlane[a]):
                l = scn.agent_lanes[a, k]
                queues.append(self.ms.lane_stats(l, float(scn.lane_origin[l]))[1])     # lane.getLastStepHaltingNumber
        queues = np.array(queues)
        self.traffic_data.append({'episode': self.cur_episode, 'time_sec': self.cur_sec, 'number_total_car': n,
                                  'number_departed_car': tot['step_departed'], 'number_arrived_car': tot['step_arrived'],
                                  'avg_wait_sec': wsum / n if n > 0 else 0, 'avg_speed_mps': ssum / n if n > 0 else 0,
                                  'std_queue': np.std(queues), 'avg_queue': np.mean(queues)})

    def collect_tripinfo(self):                            # env.py:498-515 (the tripinfo file SUMO writes at close)
        for r, ser, dep, arr, wsec, wcnt in self.ms.trips():
            if arr < 0:                                        # truncated by the teleport surrogate: not in SUMO's tripinfo file
                self.teleported_trips = getattr(self, 'teleported_trips', 0) + 1
                if not hasattr(self, 'truncated_trip_data'):
                    self.truncated_trip_data = []
                self.truncated_trip_data.append({'episode': self.cur_episode, 'id': 'f_%d.%d' % (r, ser), 'depart_sec': '%.2f' % dep,
                                                 'removed_sec': '%.2f' % (-arr), 'duration_sec': '%.2f' % (-arr - dep),
                                                 'wait_step': '%d' % wcnt, 'wait_sec': '%.2f' % wsec})
                continue
            self.trip_data.append({'episode': self.cur_episode, 'id': 'f_%d.%d' % (r, ser), 'depart_sec': '%.2f' % dep,
                                   'arrival_sec': '%.2f' % arr, 'duration_sec': '%.2f' % (arr - dep), 'wait_step': '%d' % wcnt,
                                   'wait_sec': '%.2f' % wsec})

    def update_fingerprint(self, policy):                  # env.py:633-635
        self.fingerprint = [np.array(pi)[:-1] for pi in policy]

    def reset(self, test_ind=0):                           # env.py:544-561
        self.prev_action = [0] * self.n_agent              # env.py:448
        seed = self.seed if self.train_mode else self.test_seeds[test_ind]
        self.ms.reset(seed, episode_stream_routes(self.scn, seed))
        self.cur_sec = 0
        self.cur_episode += 1
        if self.agent == 'ma2c':
            self.update_fingerprint([np.array([1. / n] * n) for n in self.n_a_ls])   # env.py:263-269
        self.seed += 1
        wave, wait, _ = self._measure()
        return self._get_state(wave, wait)

    def step(self, action):                                # env.py:566-631
        scn = self.scn
        self._set_phase(action, 'yellow')
        self._simulate(scn.yellow_interval_sec)
        rest = scn.control_interval_sec - scn.yellow_interval_sec
        self._set_phase(action, 'green')
        self._simulate(rest)
        wave, wait, halt = self._measure()
        state = self._get_state(wave, wait)
        reward = self._reward(wait, halt)
        done = self.cur_sec >= scn.episode_length_sec
        global_reward = np.sum(reward)
        if self.is_record:                                 # env.py:581-588
            self.control_data.append({'episode': self.cur_episode, 'time_sec': self.cur_sec,
                                      'step': self.cur_sec / scn.control_interval_sec,
                                      'action': ','.join(['%d' % a for a in action]), 'reward': global_reward})
        if not self.train_mode:
            return state, reward, done, global_reward
        if self.agent in ('a2c', 'greedy'):
            reward = global_reward
        elif self.agent != 'ma2c':
            reward = np.array([global_reward] * len(reward))
            if scn.reward_scale_realnet:
                reward = reward / (self.n_agent * REALNET_REWARD_NORM)
        else:
            new_reward = []
            for a, r in enumerate(reward):
                cur = r
                for j in scn.neighbors[a]:
                    cur += scn.coop_gamma * reward[j]
                if not scn.reward_scale_realnet:
                    new_reward.append(cur)
                else:
                    new_reward.append(cur / ((1 + len(scn.neighbors[a])) * REALNET_REWARD_NORM))
            reward = np.array(new_reward)
        return state, reward, done, global_reward


def greedy_large_grid(ob):
    """envs/large_grid_env.py:56-60 -- deterministic action source for tests/bench."""
    flows = [ob[0] + ob[3], ob[2] + ob[5], ob[1] + ob[4], ob[1] + ob[2], ob[4] + ob[5]]
    return int(np.argmax(np.array(flows)))
