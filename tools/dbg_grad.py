import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tests.test_model_gpu as tm
from deeprl_signal_control_amd import _lib
agent, E, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
use_cache = int(sys.argv[4]); seed = int(sys.argv[5]) if len(sys.argv) > 5 else E * T
scn, m, o = tm._make(agent, E, T, seed=5)
rng = np.random.RandomState(seed)
m.reset(); o.reset()
obs, done = tm._fill(scn, m, o, E, T, rng, use_cache=bool(use_cache))
Rb = m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone()
_lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(Rb.data_ptr()), 0.01))
ograds, _ = o.compute_grads(Rb.cpu().numpy(), 0.01)
g = m.unpack(m.grad_tensor().cpu().numpy())
worst = {}
for t in range(m.G):
    for k, og in ograds[t].items():
        og = og.numpy(); sc = max(np.abs(og).max(), 1e-7)
        e = np.abs(g[t][k] - og).max() / sc
        if e > worst.get(k, (0, 0))[0]: worst[k] = (e, t)
print(os.environ.get('TSC_UNFUSED_DX'), os.environ.get('TSC_UNFUSED_DW'), 'cache', use_cache, {k: ('%.1e' % v[0], v[1]) for k, v in worst.items()})
