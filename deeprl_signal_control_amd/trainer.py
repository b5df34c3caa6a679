"""Host loop mirroring the reference's Trainer (utils.py:110-308) for E parallel env instances.

``VecTrainer.explore`` is ``Trainer.explore`` (utils.py:142-193) with every per-agent Python loop
replaced by one batched call; ``run_iteration`` = explore + ``model.backward`` (utils.py:284-295).
All env instances run in lock-step, so every episode ends on the same control step (T % n_step
== 0, utils.py:121) and the whole batch is reset together (utils.py:277-283).
"""
import numpy as np
import torch


class Counter:
    """utils.py:70-107."""

    def __init__(self, total_step, test_step, log_step):
        self.cur_step, self.cur_test_step = 0, 0
        self.total_step, self.test_step, self.log_step, self.stop = total_step, test_step, log_step, False

    def next(self):
        self.cur_step += 1
        return self.cur_step

    def should_test(self):
        if (self.cur_step - self.cur_test_step) >= self.test_step:
            self.cur_test_step = self.cur_step
            return True
        return False

    def should_log(self):
        return self.cur_step % self.log_step == 0

    def should_stop(self):
        return self.cur_step >= self.total_step or self.stop


class VecTrainer:
    def __init__(self, env, model, global_counter=None, log_rewards=False):
        self.env, self.model = env, model
        self.log_rewards = log_rewards                          # keep the per-step global rewards of the episode (utils.py:161)
        self._ep_rewards = []
        self.agent = env.agent
        self.n_step = model.n_step
        assert env.T % self.n_step == 0                         # utils.py:121
        self.global_counter = global_counter or Counter(1 << 62, 1 << 62, 1 << 62)
        self.ob = None
        self.done = True
        self.episode_step = 0
        self.zero_R = torch.zeros(env.E, env.A, dtype=torch.float32, device=env.device)
        # zero-copy rollouts: forward and env write the transition straight into the learner's on-policy buffer
        self.slots = model.rollout_slots() if (self.agent.endswith('a2c') and hasattr(model, 'rollout_slots')) else None

    def start_episode(self):
        """utils.py:277-283: env.reset(); done = True (pre-decision, resets LSTM state); model.reset()."""
        if self.slots is not None:
            self.ob = self.env.reset(obs_out=self.slots['obs'][0])
            self.slots['done'][0].fill_(1)                        # pre-decision done of the first step
            self.model.cur_t = 0
        else:
            self.ob = self.env.reset()
        self.done = True
        self.model.reset()
        self.episode_step = 0

    def _explore_q(self):
        """utils.py:142-193 for the value-based agents (IQL): epsilon-greedy forward, replay transitions, R = 0."""
        env, model = self.env, self.model
        ob = self.ob
        finished = False
        for _ in range(self.n_step):
            action, _ = model.forward(ob, mode='explore')                    # utils.py:159
            next_ob, reward, done_post, g = env.step(action)
            if self.log_rewards:
                self._ep_rewards.append(g.clone())
            self.global_counter.next()
            self.episode_step += 1
            model.add_transition(ob, action, reward, next_ob, done_post)    # utils.py:166-167
            finished = env.cur_sec >= env.scn.episode_length_sec
            ob = next_ob
            if finished:
                break
        self.ob, self.done = ob, done_post
        return finished, None

    def _explore_slots(self):
        """explore() with the transition written in place: slot t of the on-policy buffer receives the action and value
        (by the fused forward), the raw reward and the post-step done (by the env), slot t + 1 the next observation -- no
        add_transition copy (agents/models.py:222-229 reduces to the reward normalisation, done when the returns are taken)."""
        env, model, sl = self.env, self.model, self.slots
        finished = False
        assert model.cur_t == 0
        for t in range(self.n_step):
            pi, v, action = model.forward_sample(sl['obs'][t], sl['done'][t], v_out=sl['value'][t], action_out=sl['action'][t])
            if self.agent == 'ma2c':
                env.update_fingerprint(pi, zero_copy=True)
            _, _, _, global_reward = env.step(action, obs_out=sl['obs'][t + 1], reward_out=sl['reward'][t], done_out=sl['done'][t + 1])
            if self.log_rewards:
                self._ep_rewards.append(global_reward.clone())
            self.global_counter.next()
            self.episode_step += 1
            model.commit_transition()
            finished = env.cur_sec >= env.scn.episode_length_sec
            if finished:
                break
        assert model.cur_t == self.n_step, 'episodes end on a rollout boundary (T %% n_step == 0, utils.py:121)'
        self.ob, self.done = sl['obs'][self.n_step], sl['done'][self.n_step]
        R = self.zero_R if finished else model.forward(self.ob, False, 'v')
        # after backward() the library copies slot n_step -> slot 0 (obs) and done[n_step] -> done[0]
        return finished, R

    def explore(self):
        """utils.py:142-193.  Returns (episode_finished, R) with R the bootstrap values."""
        if not self.agent.endswith('a2c'):
            return self._explore_q()
        if self.slots is not None:
            return self._explore_slots()
        env, model = self.env, self.model
        ob, done = self.ob, self.done
        finished = False
        for _ in range(self.n_step):
            pi, v, action = model.forward_sample(ob, done)      # forward + np.random.choice (utils.py:148-157)
            if self.agent == 'ma2c':
                env.update_fingerprint(pi, zero_copy=True)       # before step (utils.py:149-151); pi is not
                                                                 # touched again until the next forward
            next_ob, reward, done_post, global_reward = env.step(action)
            if self.log_rewards:
                self._ep_rewards.append(global_reward.clone())
            self.global_counter.next()
            self.episode_step += 1
            model.add_transition(ob, done, action, reward, v, done_post)
            finished = env.cur_sec >= env.scn.episode_length_sec  # all instances share the clock
            ob, done = next_ob, done_post
            if finished:
                break
        self.ob, self.done = ob, done
        if finished:
            R = self.zero_R                                      # utils.py:187-188
        else:
            R = model.forward(ob, False, 'v')                    # utils.py:190
        return finished, R

    def run_iteration(self, want_stats=False):
        """One on-policy A2C iteration: n_step control steps of every env instance + one update."""
        if self.ob is None:
            self.start_episode()
        finished, R = self.explore()
        if self.agent.endswith('a2c'):
            stats = self.model.backward(R, want_stats=want_stats)
        else:
            stats = self.model.backward(want_stats=want_stats)         # utils.py:291
        if finished:
            self.env.terminate()                                   # utils.py:293-294
            self.start_episode()
        return finished, stats

    def run(self, total_iterations):
        for _ in range(total_iterations):
            self.run_iteration()
            if self.global_counter.should_stop():
                break

    def run_training(self, run_test=False, output_path=None, policy_type='default'):
        """Trainer.run (utils.py:255-308): episodes until the counter stops; with run_test the test block every
        test_interval control steps; one log row per training episode (test_id -1: mean / std over the episode's control
        steps of the global reward, averaged over the env instances).  Returns the rows of train_reward.csv."""
        import logging
        data = []
        was = self.log_rewards
        self.log_rewards = True
        while not self.global_counter.should_stop():
            if run_test and self.global_counter.should_test():
                rows = self.evaluate(policy_type, step=self.global_counter.cur_step)
                data += rows
                logging.info('Testing: global step %d, avg R: %.2f' % (self.global_counter.cur_step, np.mean([r['avg_reward'] for r in rows])))
            self.env.train_mode = True
            self.start_episode()
            self._ep_rewards = []
            while True:
                finished, R = self.explore()
                if self.agent.endswith('a2c'):
                    self.model.backward(R)
                else:
                    self.model.backward()
                if finished:
                    self.env.terminate()
                    break
            r = torch.stack(self._ep_rewards)                       # [T, E]
            mean, std = float(r.mean(0).mean().item()), float(r.std(0, unbiased=False).mean().item())
            data.append({'agent': self.agent, 'step': self.global_counter.cur_step, 'test_id': -1, 'avg_reward': mean, 'std_reward': std})
            logging.info('Training: global step %d, episode %d, avg R: %.2f' % (self.global_counter.cur_step, self.env.cur_episode, mean))
            self.ob = None
        self.log_rewards = was
        return data

    # ---- evaluation (utils.py:195-234, 257-275) ---------------------------------------------
    def perform(self, test_ind=0, policy_type='default'):
        """Trainer.perform for every env instance at once: one test episode in the env's current mode
        (the caller sets train_mode = False for un-shaped rewards), actions sampled from the policy
        ('default' / 'stochastic') or its argmax ('deterministic').  test_ind: one index, or one per instance.
        Returns (mean, std) of the global reward over the episode's control steps, float64 arrays [E]
        (np.mean / np.std of the reference's `rewards` list, per instance)."""
        env, model = self.env, self.model
        ob = env.reset(test_ind=test_ind)
        done = True                                               # pre-decision: resets the LSTM state
        model.reset()
        rewards = []
        while True:
            if self.agent == 'greedy':                           # utils.py:202-203
                action = model.forward(ob)
            elif not self.agent.endswith('a2c'):                 # value-based (utils.py:221-226)
                action, _ = model.forward(ob, stochastic=policy_type == 'stochastic')
            else:
                pi = model.forward(ob, done, 'p')
                if self.agent == 'ma2c':
                    env.update_fingerprint(pi)
                if policy_type != 'deterministic':
                    action = model.sample(pi)
                else:
                    action = pi.argmax(dim=-1).to(torch.int32)   # padded actions have probability 0
            ob, _, done, g = env.step(action)
            rewards.append(g.clone())
            if env.cur_sec >= env.scn.episode_length_sec:
                break
        r = torch.stack(rewards)                                  # [T, E] float64
        self.ob = None                                            # the training episode has to restart
        return r.mean(0).cpu().numpy(), r.std(0, unbiased=False).cpu().numpy()

    def evaluate(self, policy_type='default', output_path=None, step=0):
        """The test block of Trainer.run (utils.py:257-275): every test seed once, un-shaped rewards.  With
        E >= test_num all seeds run in ONE batched episode (instance e evaluates seed e % test_num).
        Returns the reference's log rows; output_path appends them to `train_reward.csv`'s schema."""
        env = self.env
        was = env.train_mode
        env.train_mode = False
        try:
            rows = []
            if env.E >= env.test_num:
                inds = np.arange(env.E) % env.test_num
                mean, std = self.perform(inds, policy_type)
                for t in range(env.test_num):
                    rows.append({'agent': self.agent, 'step': step, 'test_id': t,
                                 'avg_reward': float(mean[inds == t].mean()), 'std_reward': float(std[inds == t].mean())})
            else:
                for t in range(env.test_num):
                    mean, std = self.perform(t, policy_type)
                    env.terminate()
                    rows.append({'agent': self.agent, 'step': step, 'test_id': t,
                                 'avg_reward': float(mean.mean()), 'std_reward': float(std.mean())})
        finally:
            env.train_mode = was
        if output_path is not None:
            import csv
            import os
            path = os.path.join(output_path, 'train_reward.csv')
            new_file = not os.path.exists(path)
            with open(path, 'a', newline='') as fh:
                w = csv.DictWriter(fh, fieldnames=['agent', 'avg_reward', 'std_reward', 'step', 'test_id'])
                if new_file:
                    w.writeheader()
                w.writerows(rows)
        return rows

    def mean_step_reward(self):
        steps = max(1, self.global_counter.cur_step)
        return self.env.reward_sum() / (steps * self.env.E)


class MultiBatchTrainer:
    """The same iteration with the env instances of one GPU split into B independent half-batches, each on
    its own HIP stream with its own (env, model) handles and identical parameters.  Within a half-batch the
    rollout is a strict chain (forward -> step -> forward ...); the latency-bound microsimulator step of one
    half-batch runs under the MFMA-bound policy forward of the other.  The update is what N ranks would do
    on N GPUs: every half computes the gradient of its own samples, the flat buffers are summed, every
    handle applies the same averaged gradient (so the replicas stay bit-identical)."""

    def __init__(self, envs, models):
        assert len(envs) == len(models) and len(envs) >= 1
        self.envs, self.models = envs, models
        from .agents import replica_sample_seed
        for b, m in enumerate(models[1:], 1):
            # replicas of ONE learner: identical parameters / optimizer state, independent action streams
            m.copy_from(models[0])
            if m.sample_seed == models[0].sample_seed:
                m.sample_seed = replica_sample_seed(models[0].sample_seed, models[0].rank, b)
        self.parts = [VecTrainer(e, m) for e, m in zip(envs, models)]
        dev = envs[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in envs]
        for e, m, st in zip(envs, models, self.streams):
            e.use_stream(st)
            m.use_stream(st)
        self.n_step = models[0].n_step

    def _each(self, fn):
        out = []
        for part, st in zip(self.parts, self.streams):
            with torch.cuda.stream(st):
                out.append(fn(part))
        return out

    def run_iteration(self):
        for part in self.parts:
            if part.ob is None:
                with torch.cuda.stream(self.streams[self.parts.index(part)]):
                    part.start_episode()
        # rollout: interleave the half-batches control step by control step so both chains stay fed
        state = [dict(ob=p.ob, done=p.done, fin=False) for p in self.parts]
        for _ in range(self.n_step):
            for p, st, s in zip(self.parts, self.streams, state):
                if s['fin']:
                    continue
                with torch.cuda.stream(st):
                    env, model = p.env, p.model
                    pi, v, action = model.forward_sample(s['ob'], s['done'])
                    if p.agent == 'ma2c':
                        env.update_fingerprint(pi, zero_copy=True)
                    next_ob, reward, done_post, _ = env.step(action)
                    p.global_counter.next()
                    model.add_transition(s['ob'], s['done'], action, reward, v, done_post)
                    s['fin'] = env.cur_sec >= env.scn.episode_length_sec
                    s['ob'], s['done'] = next_ob, done_post
        finished = state[0]['fin']
        for p, st, s in zip(self.parts, self.streams, state):
            with torch.cuda.stream(st):
                p.ob, p.done = s['ob'], s['done']
                R = p.zero_R if s['fin'] else p.model.forward(s['ob'], False, 'v')
                p.model.compute_grads(R)
        # gradient exchange between the half-batches (and, with several GPUs, between ranks)
        g0 = self.models[0].grad_tensor()
        main = self.streams[0]
        for m, st in zip(self.models[1:], self.streams[1:]):
            main.wait_stream(st)
        with torch.cuda.stream(main):
            for m in self.models[1:]:
                g0.add_(m.grad_tensor())
            scale = 1.0 / len(self.models)
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                from .agents import allreduce_grads_
                scale *= allreduce_grads_(g0, self.models[0].pg)
            for m in self.models[1:]:
                m.grad_tensor().copy_(g0)
        for st in self.streams[1:]:
            st.wait_stream(main)
        self._each(lambda p: p.model.apply_grads(scale))
        if finished:
            def restart(p):
                p.env.terminate()
                p.start_episode()
            self._each(restart)
        return finished, None

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def mean_step_reward(self):
        tot = sum(p.env.reward_sum() for p in self.parts)
        steps = max(1, self.parts[0].global_counter.cur_step)
        return tot / (steps * sum(p.env.E for p in self.parts))


def greedy_actions(scn, wave):
    """HOST restatement of the reference's greedy controllers (envs/large_grid_env.py:56-60,
    envs/real_net_env.py:90-111, envs/small_grid_env.py:51-55) over Scenario.greedy_controller_tables -- the same tables the
    device kernel reads (VecTrafficEnv.greedy_actions -> tsc_env_greedy_actions, the product path for device tensors).  For
    host arrays only: CPU tools and tests that drive the oracle.  wave [..., A, >= terms]: the agents' own wave entries
    first (as in every obs layout), float64 like the controllers' input; candidates are summed in table order from 0 and
    np.argmax keeps the first maximum."""
    if torch.is_tensor(wave):
        raise TypeError('greedy_actions is the host restatement; device tensors go through VecTrafficEnv.greedy_actions (HIP)')
    n_cand, term, action = scn.greedy_controller_tables()
    w = np.asarray(wave, np.float64)
    out = np.zeros(w.shape[:-1], np.int32)
    for a in range(scn.n_agent):
        flows = []
        for c in range(int(n_cand[a])):
            f = np.zeros(w.shape[:-2], np.float64)
            for j in term[a, c]:
                if j < 0:
                    break
                f = f + w[..., a, int(j)]
            flows.append(f)
        out[..., a] = action[a][np.argmax(np.stack(flows, -1), -1)]
    return out
