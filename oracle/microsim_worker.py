"""One worker of bench.py's `cpu_baseline.sim_only_all_cores` leg (SURVEY.md 8d baseline ii): the C microsim alone
(oracle/microsim.c; no env wrapper, no nets) over `n_inst` env instances of large_grid, one full episode each, under a fixed
30-second signal cycle.  TEST / MEASUREMENT INFRASTRUCTURE ONLY -- see oracle/__init__.py.

    python -m oracle.microsim_worker N_INST SEED0        (prints "ready", waits for a line on stdin, prints its wall time)

The parent (bench.py) starts one worker per core, releases them together and divides the env-steps of all workers by the wall
time from the release to the last answer."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    n_inst, seed0 = int(sys.argv[1]), int(sys.argv[2])
    from deeprl_signal_control_amd.scenario import build_large_grid
    from oracle.microsim import MicroSim
    scn = build_large_grid('ma2c')
    sims = [MicroSim(scn) for _ in range(n_inst)]
    print('ready', flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    live = 0
    for i, ms in enumerate(sims):
        ms.reset(seed0 + i)
        for t in range(0, scn.episode_length_sec, scn.control_interval_sec):
            for a in range(scn.n_agent):
                ms.set_links(a, scn.phases[a][(t // 30) % 5])
            ms.step(scn.control_interval_sec)
            live += ms.totals()['live']
    dt = time.perf_counter() - t0
    print('%.6f %.3f' % (dt, live / (n_inst * (scn.episode_length_sec // scn.control_interval_sec))), flush=True)


if __name__ == '__main__':
    main()
