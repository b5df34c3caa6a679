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
    def __init__(self, env, model, global_counter=None):
        self.env, self.model = env, model
        self.agent = env.agent
        self.n_step = model.n_step
        assert env.T % self.n_step == 0                         # utils.py:121
        self.global_counter = global_counter or Counter(1 << 62, 1 << 62, 1 << 62)
        self.ob = None
        self.done = True
        self.episode_step = 0
        self.zero_R = torch.zeros(env.E, env.A, dtype=torch.float32, device=env.device)

    def start_episode(self):
        """utils.py:277-283: env.reset(); done = True (pre-decision, resets LSTM state); model.reset()."""
        self.ob = self.env.reset()
        self.done = True
        self.model.reset()
        self.episode_step = 0

    def explore(self):
        """utils.py:142-193.  Returns (episode_finished, R) with R the bootstrap values."""
        env, model = self.env, self.model
        ob, done = self.ob, self.done
        finished = False
        for _ in range(self.n_step):
            pi, v, action = model.forward_sample(ob, done)      # forward + np.random.choice (utils.py:148-157)
            if self.agent == 'ma2c':
                env.update_fingerprint(pi, zero_copy=True)       # before step (utils.py:149-151); pi is not
                                                                 # touched again until the next forward
            next_ob, reward, done_post, global_reward = env.step(action)
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
        stats = self.model.backward(R, want_stats=want_stats)
        if finished:
            self.env.terminate()                                   # utils.py:293-294
            self.start_episode()
        return finished, stats

    def run(self, total_iterations):
        for _ in range(total_iterations):
            self.run_iteration()
            if self.global_counter.should_stop():
                break

    def mean_step_reward(self):
        steps = max(1, self.global_counter.cur_step)
        return self.env.reward_sum() / (steps * self.env.E)


def greedy_actions_large_grid(obs):
    """envs/large_grid_env.py:56-60 on the batched obs tensor [E,A,SMAX] (first 6 entries = own
    wave).  Host-side helper for sim-only benchmarks; plain indexing, no learned compute."""
    w = obs[..., :6]
    flows = torch.stack([w[..., 0] + w[..., 3], w[..., 2] + w[..., 5], w[..., 1] + w[..., 4],
                         w[..., 1] + w[..., 2], w[..., 4] + w[..., 5]], -1)
    return flows.argmax(-1).to(torch.int32)
