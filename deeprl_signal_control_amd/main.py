"""Command line of the reference (main.py:21-48,82-230) on the MI355X path.

    python -m deeprl_signal_control_amd.main --base-dir DIR train --config-dir config/config_ma2c_large.ini
                                                                  [--test-mode no_test|in_train_test|after_train_test|all_test]
                                                                  [--envs E]
    python -m deeprl_signal_control_amd.main --base-dir DIR evaluate --agents ma2c,greedy
                                                                  [--evaluation-policy-type default|stochastic|deterministic]
                                                                  [--evaluation-seeds 10000,20000,...]

Same sub-commands, flags, INI sections ([MODEL_CONFIG] [TRAIN_CONFIG] [ENV_CONFIG], config/config_*.ini of the
reference are read unchanged) and on-disk layout as the reference: ``DIR/{log,data,model}`` with the config copied
into ``data/`` (main.py:84-87), ``data/train_reward.csv`` (utils.py:299-308), ``model/checkpoint-<step>``
(agents/models.py:83-108; an .npz here), and for ``evaluate``: ``DIR/<agent>/{data,model}`` in, ``DIR/eva_data/
<scenario>_<agent>_{control,traffic,trip}.csv`` out (main.py:158-222, utils.py:366-388, envs/env.py:534-542).

What differs: ``--envs E`` trains on E parallel env instances per GPU (the reference has one); `total_step`,
`test_interval`, `log_interval` keep counting control steps of ONE instance, so a run is E times the experience.
The evaluation runs all evaluation seeds as one batched episode.  TensorBoard summaries and ``--demo`` (SUMO gui) have no
equivalent here.
"""
import argparse
import configparser
import logging
import os
import shutil
import sys
import time

import numpy as np


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('--base-dir', type=str, required=False, default='./signal_control_results', help='experiment base dir')
    subparsers = parser.add_subparsers(dest='option', help='train or evaluate')
    sp = subparsers.add_parser('train', help='train a single agent under base dir')
    sp.add_argument('--test-mode', type=str, required=False, default='no_test',
                    choices=['no_test', 'in_train_test', 'after_train_test', 'all_test'], help='test mode during training')
    sp.add_argument('--config-dir', type=str, required=False, default='./config/config_ma2c_large.ini', help='experiment config path')
    sp.add_argument('--envs', type=int, default=1, help='parallel env instances on this GPU (the reference: 1)')
    sp.add_argument('--device', type=int, default=0)
    sp = subparsers.add_parser('evaluate', help='evaluate and compare agents under base dir')
    sp.add_argument('--agents', type=str, required=False, default='naive', help='agent folder names for evaluation, split by ,')
    sp.add_argument('--evaluation-policy-type', type=str, required=False, default='default',
                    help='inference policy type in evaluation: default, stochastic, or deterministic')
    sp.add_argument('--evaluation-seeds', type=str, required=False, default=','.join([str(i) for i in range(10000, 100001, 10000)]),
                    help='random seeds for evaluation, split by ,')
    sp.add_argument('--demo', action='store_true', help='accepted for compatibility (there is no gui)')
    sp.add_argument('--device', type=int, default=0)
    args = parser.parse_args(argv)
    if not args.option:
        parser.print_help()
        sys.exit(1)
    return args


# ---- utils.py:19-67 ---------------------------------------------------------------------------------------------
def init_dir(base_dir, pathes=('log', 'data', 'model')):
    os.makedirs(base_dir, exist_ok=True)
    dirs = {}
    for path in pathes:
        cur = base_dir + '/%s/' % path
        os.makedirs(cur, exist_ok=True)
        dirs[path] = cur
    return dirs


def init_log(log_dir):
    logging.basicConfig(format='%(asctime)s [%(levelname)s] %(message)s', level=logging.INFO, force=True,
                        handlers=[logging.FileHandler('%s/%d.log' % (log_dir, time.time())), logging.StreamHandler()])


def init_test_flag(test_mode):
    return {'no_test': (False, False), 'in_train_test': (True, False), 'after_train_test': (False, True),
            'all_test': (True, True)}[test_mode]


def find_file(cur_dir, suffix='.ini'):
    for f in sorted(os.listdir(cur_dir)):
        if f.endswith(suffix):
            return cur_dir + '/' + f
    logging.error('Cannot find %s file' % suffix)
    return None


class GreedyPolicy:
    """LargeGridController / RealNetController / SmallGridController (envs/*_env.py) on the device obs tensor: the env's
    greedy kernel (VecTrafficEnv.greedy_actions -> tsc_env_greedy_actions) over the scenario's controller tables."""
    name = 'greedy'
    n_step = 1

    def __init__(self, env):
        self.env = env

    def forward(self, ob, *_a, **_k):
        return self.env.greedy_actions(ob)

    def reset(self):
        pass


def init_model(env, config, total_step, n_env, seed, device=0):
    """main.py:102-118: the learner for env.agent."""
    from .agents import VecA2C
    from .iql import VecIQL
    a_max = int(env.scn.green_tab.shape[1])
    if env.agent in ('ia2c', 'ma2c'):
        return VecA2C(env.n_s_ls, env.n_a_ls, env.n_w_ls, env.n_f_ls, n_env, env.scn.s_max, a_max, config['MODEL_CONFIG'],
                      total_step, device=device, seed=seed, name=env.agent)
    if env.agent in ('iqld', 'iqll'):
        return VecIQL(env.n_s_ls, env.n_a_ls, env.n_w_ls, n_env, env.scn.s_max, a_max, config['MODEL_CONFIG'], total_step,
                      device=device, seed=0, model_type='dqn' if env.agent == 'iqld' else 'lr')
    raise ValueError('agent %r has no learner (main.py:102-118 knows ia2c, ma2c, iqld, iqll)' % env.agent)


def train(args):
    """main.py:82-155 + utils.py:255-308 (Trainer.run)."""
    from .env import VecTrafficEnv, scenario_from_config
    from .trainer import Counter, VecTrainer
    dirs = init_dir(args.base_dir)
    init_log(dirs['log'])
    shutil.copy(args.config_dir, dirs['data'])
    config = configparser.ConfigParser()
    config.read(args.config_dir)
    in_test, post_test = init_test_flag(args.test_mode)
    scn, seed, test_seeds = scenario_from_config(config['ENV_CONFIG'])
    env = VecTrafficEnv(scn, args.envs, device=args.device, seed=seed, test_seeds=test_seeds)
    logging.info('Training: s dim: %d, a dim %d, s dim ls: %r, a dim ls: %r' % (env.n_s, env.n_a, env.n_s_ls, env.n_a_ls))
    total_step = int(config.getfloat('TRAIN_CONFIG', 'total_step'))
    test_step = int(config.getfloat('TRAIN_CONFIG', 'test_interval'))
    log_step = int(config.getfloat('TRAIN_CONFIG', 'log_interval'))
    counter = Counter(total_step, test_step, log_step)
    model = init_model(env, config, total_step, args.envs, seed, args.device)
    trainer = VecTrainer(env, model, counter, log_rewards=True)
    data = trainer.run_training(run_test=in_test, output_path=dirs['data'])
    if post_test:                                               # Tester.run_offline (utils.py:324-338)
        rows = trainer.evaluate('default', step=counter.cur_step)
        data += rows
        logging.info('Offline testing: avg R: %.2f' % np.mean([r['avg_reward'] for r in rows]))
    write_reward_csv(data, dirs['data'] + 'train_reward.csv')
    logging.info('Training: save final model at step %d ...' % counter.cur_step)
    model.save(dirs['model'], counter.cur_step)
    env.close(); model.close()
    return data


def write_reward_csv(rows, path):
    """utils.py:307-308: pd.DataFrame(self.data).to_csv(...) (columns in pandas' alphabetical order of that era)."""
    import pandas as pd
    df = pd.DataFrame(rows)
    if len(df.columns):
        df = df[sorted(df.columns)]
    df.to_csv(path)


def evaluate_agent(agent_dir, output_dir, seeds, policy_type='default', device=0):
    """main.py:158-198 + Evaluator.run (utils.py:366-388): all evaluation seeds as ONE batched, recorded episode."""
    from .env import VecTrafficEnv, scenario_from_config
    from .trainer import VecTrainer
    agent = agent_dir.rstrip('/').split('/')[-1]
    if not os.path.isdir(agent_dir):
        logging.error('Evaluation: %s does not exist!' % agent)
        return None
    config_dir = find_file(agent_dir + '/data/')
    if not config_dir:
        return None
    config = configparser.ConfigParser()
    config.read(config_dir)
    if agent == 'greedy':
        config['ENV_CONFIG']['agent'] = 'greedy'
    scn, seed, _ = scenario_from_config(config['ENV_CONFIG'])
    E = len(seeds)
    env = VecTrafficEnv(scn, E, device=device, seed=seed, test_seeds=seeds)
    logging.info('Evaluation: s dim: %d, a dim %d, s dim ls: %r, a dim ls: %r' % (env.n_s, env.n_a, env.n_s_ls, env.n_a_ls))
    if agent != 'greedy':
        model = init_model(env, config, 0, E, seed, device)
        if not model.load(agent_dir + '/model/'):
            logging.error('Evaluation: no checkpoint under %s/model/' % agent_dir)
            return None
    else:
        model = GreedyPolicy(env)
    env.train_mode = False
    env.set_record(True)
    trainer = VecTrainer(env, model)
    mean, std = trainer.perform(np.arange(E), policy_type)
    env.collect_tripinfo()
    for e in range(E):
        logging.info('test %i, avg reward %.2f' % (e, mean[e]))
    write_eval_tables(env, output_dir)
    env.close()
    if hasattr(model, 'close'):
        model.close()
    return mean, std


def write_eval_tables(env, output_dir):
    """envs/env.py:534-542 over all evaluated instances: instance e is episode e + 1 (the reference runs the seeds one
    after the other and numbers them by cur_episode)."""
    import pandas as pd
    for kind, per_env in (('control', env.control_data), ('traffic', env.traffic_data), ('trip', env.trip_data),
                          ('trip_truncated', getattr(env, 'truncated_trip_data', []))):
        rows = []
        for e, rs in enumerate(per_env):
            rows += [dict(r, episode=e + 1) for r in rs]
        if kind == 'trip_truncated':
            if not rows:
                continue
            # trips the teleport surrogate cut short (MICROSIM_SPEC.md rule 1): not in the trip table (SUMO's tripinfo would list them
            # later, with long durations), so averages over the trip table alone are biased low -- say so where it is read
            logging.info('Evaluation: %d trips truncated by the teleport surrogate (mean %.1f s in the network, %.1f s waiting) are in '
                         '%s_%s_trip_truncated.csv, not in the trip table' % (
                             len(rows), np.mean([float(r['duration_sec']) for r in rows]), np.mean([float(r['wait_sec']) for r in rows]),
                             env.scn.name, env.agent))
        df = pd.DataFrame(rows)
        if len(df.columns):
            df = df[sorted(df.columns)]
        df.to_csv(output_dir + ('%s_%s_%s.csv' % (env.scn.name, env.agent, kind)))


def evaluate(args):
    """main.py:201-222 (agents one after the other instead of one thread + SUMO port each)."""
    dirs = init_dir(args.base_dir, pathes=['eva_data', 'eva_log'])
    init_log(dirs['eva_log'])
    seeds = [int(s) for s in args.evaluation_seeds.split(',')] if args.evaluation_seeds else []
    logging.info('Evaluation: policy type: %s, random seeds: %s' % (args.evaluation_policy_type, seeds))
    out = {}
    for agent in args.agents.split(','):
        out[agent] = evaluate_agent(args.base_dir + '/' + agent, dirs['eva_data'], seeds, args.evaluation_policy_type, args.device)
    return out


def main(argv=None):
    args = parse_args(argv)
    return train(args) if args.option == 'train' else evaluate(args)


if __name__ == '__main__':
    main()
