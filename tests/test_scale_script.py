"""tools/scale.sh -- the one-command 1 -> 8 GPU curve: its argument handling and the launch lines it would
issue, checked here without a GPU (--dry-run); the curve itself needs the 8-GPU node the driver owns."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SH = os.path.join(ROOT, 'tools', 'scale.sh')


def _run(*args):
    return subprocess.run(['bash', SH] + list(args), capture_output=True, text=True, cwd=ROOT)


def test_default_curve_is_c3_and_c5_over_1_2_4_8(tmp_path):
    r = _run('--dry-run', '--out', str(tmp_path))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 8
    for i, (c, n) in enumerate([(c, n) for c in ('c3', 'c5') for n in (1, 2, 4, 8)]):
        ln = lines[i]
        assert '--config %s --gpus %d ' % (c, n) in ln and ln.endswith('scale_%s_%d.json' % (c, n))
        if n == 1:
            assert 'torch.distributed.run' not in ln
        else:       # the driver's own launch line (one rank per GPU over RCCL, rendezvous on 127.0.0.1)
            assert '-m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port 29533 ' % n in ln
        assert '--backend' not in ln            # bench.py's default backend is nccl; the script never selects gloo


def test_options_and_rejections(tmp_path):
    r = _run('--dry-run', '--configs', 'c3', '--gpus', '2 8', '--steps', '6', '--warmup', '12', '--port', '29999', '--out', str(tmp_path))
    assert r.returncode == 0 and len(r.stdout.strip().splitlines()) == 2
    assert all('--steps 6 --warmup 12' in ln and '--master-port 29999' in ln for ln in r.stdout.strip().splitlines())
    for bad in (['--gpus', '3'], ['--configs', 'c9'], ['--frobnicate']):
        r = _run('--dry-run', *bad)
        assert r.returncode == 2 and 'scale.sh' in r.stderr
