"""The HIP path LEARNS: large_grid MA2C at the benchmark batch (E = 1024), Trainer.run's episode loop
(utils.py:255-308) through tools/learning_curve.py.  The committed curve (profiles/r03_learning_curve.json, 500 episodes at
the reference's lr 5e-4: -503 -> -175 mean step reward) takes two GPU-minutes; this test runs 18 episodes (108 updates)
at lr 5e-3 -- the reference's lr belongs to a batch of 120 samples per update, here an update averages 122 880 -- and asserts
the trend: the mean step reward of the last four episodes exceeds the first four by a clear margin (measured +7 (+14 under the round-2 simulator spec) against an
episode-to-episode noise of about 1)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ma2c_mean_step_reward_improves():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import learning_curve
    rows, _ = learning_curve.run(18, 1024, lr=5e-3)
    r = np.array([x['avg_reward'] for x in rows])
    assert np.isfinite(r).all() and r[0] < -400                      # an untrained policy is about as good as a random one
    assert r[-4:].mean() > r[:4].mean() + 4.0, r
    assert np.all(np.diff(r[6:]) > -3.0), r                          # and it keeps improving, not oscillating
