"""Launcher with the reference's command-line surface (main.py:22-39) driving the MI355X agents.

    python -m rlrep_amd.main --alg sac --env Pendulum-v1 --max_timesteps 2000 --start_timesteps 500

Same flags, same per-algorithm constructor overrides (main.py:81-104), same loop structure (random actions for
`start_timesteps`, then epsilon-greedy 0.01 around `select_action(explore=True)`, one `agent.train()` per
environment step, evaluation every `eval_freq` steps).  Metrics go to `log/<env>/<alg>/<dir>/<seed>/metrics.jsonl`
as {"step": t, "info/<key>": value} lines (and to tensorboardX with the reference's tags if it is installed).
"""
import argparse
import json
import os

import numpy as np
import torch

from rlrep_amd import envs
from rlrep_amd.utils import util, buffer

EPS_GREEDY = 0.01


def build_agent(args, state_dim, action_dim, action_space):
    common = dict(state_dim=state_dim, action_dim=action_dim, action_space=action_space, discount=args.discount,
                  tau=args.tau, hidden_dim=args.hidden_dim, max_batch=args.batch_size)
    if args.alg == 'sac':
        from rlrep_amd.agent.sac.sac_agent import SACAgent
        return SACAgent(**common)
    if args.alg == 'vlsac':
        from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
        return VLSACAgent(**common, extra_feature_steps=args.extra_feature_steps, feature_dim=args.feature_dim)
    if args.alg == 'ctrlsac':
        from rlrep_amd.agent.ctrlsac.ctrlsac_agent import CTRLSACAgent
        common.update(feature_dim=2048, hidden_dim=1024)                       # hard-coded at main.py:90-91
        return CTRLSACAgent(**common, extra_feature_steps=args.extra_feature_steps)
    if args.alg == 'diffsrsac':
        from rlrep_amd.agent.diffsrsac.diffsrsac_agent import DIFFSRSACAgent
        return DIFFSRSACAgent(**common)
    if args.alg == 'spedersac':
        from rlrep_amd.agent.spedersac.spedersac_agent import SPEDERSACAgent
        return SPEDERSACAgent(**common, extra_feature_steps=5, phi_and_mu_lr=1e-5, phi_hidden_dim=512,
                              phi_hidden_depth=1, mu_hidden_dim=512, mu_hidden_depth=0,
                              critic_and_actor_lr=3e-4, critic_and_actor_hidden_dim=256)   # main.py:95-103
    raise SystemExit(f'--alg {args.alg}: not part of the MI355X hot path (see DESIGN.md, out of scope)')


def run(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--dir', default=0, type=int)
    p.add_argument('--alg', default='diffsrsac')
    p.add_argument('--env', default='HalfCheetah-v4')
    p.add_argument('--seed', default=0, type=int)
    p.add_argument('--start_timesteps', default=25e3, type=float)
    p.add_argument('--eval_freq', default=5e3, type=int)
    p.add_argument('--max_timesteps', default=1e6, type=float)
    p.add_argument('--expl_noise', default=0.1)
    p.add_argument('--batch_size', default=256, type=int)
    p.add_argument('--hidden_dim', default=256, type=int)
    p.add_argument('--feature_dim', default=256, type=int)
    p.add_argument('--discount', default=0.99)
    p.add_argument('--tau', default=0.005)
    p.add_argument('--learn_bonus', action='store_true')
    p.add_argument('--save_model', action='store_true')
    p.add_argument('--extra_feature_steps', default=3, type=int)
    p.add_argument('--eval_episodes', default=10, type=int)
    p.add_argument('--log_root', default='log')
    args = p.parse_args(argv)

    env, eval_env = envs.make(args.env), envs.make(args.env)
    env.seed(args.seed)
    eval_env.seed(args.seed)
    max_length = env._max_episode_steps
    log_path = os.path.join(args.log_root, args.env, args.alg, str(args.dir), str(args.seed))
    os.makedirs(log_path, exist_ok=True)
    jsonl = open(os.path.join(log_path, 'metrics.jsonl'), 'a')
    try:
        from tensorboardX import SummaryWriter
        tb = SummaryWriter(log_path)
    except ImportError:
        tb = None
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)

    state_dim, action_dim = env.observation_space.shape[0], env.action_space.shape[0]
    agent = build_agent(args, state_dim, action_dim, env.action_space)
    replay = buffer.ReplayBuffer(state_dim, action_dim, max_size=int(min(args.max_timesteps, 1e6)))
    evaluations = [util.eval_policy(agent, eval_env, args.eval_episodes)]

    state, done = env.reset(), False
    ep_reward, ep_steps, ep_num, info = 0.0, 0, 0, None
    timer = util.Timer()
    for t in range(int(args.max_timesteps)):
        ep_steps += 1
        if t < args.start_timesteps or np.random.uniform(0, 1) < EPS_GREEDY:
            action = env.action_space.sample()
        else:
            action = agent.select_action(state, explore=True)
        next_state, reward, done, _ = env.step(action)
        replay.add(state, action, next_state, reward, float(done) if ep_steps < max_length else 0)
        state = next_state
        ep_reward += reward
        if t >= args.start_timesteps:
            info = agent.train(replay, batch_size=args.batch_size)
        if done:
            print(f'Total T: {t + 1} Episode Num: {ep_num + 1} Episode T: {ep_steps} Reward: {ep_reward:.3f}')
            state, done = env.reset(), False
            ep_reward, ep_steps, ep_num = 0.0, 0, ep_num + 1
        if (t + 1) % args.eval_freq == 0:
            sps = timer.steps_per_sec(t + 1)
            evaluations.append(util.eval_policy(agent, eval_env, args.eval_episodes))
            if info is not None:
                row = {'step': t + 1, 'info/evaluation': float(evaluations[-1]), 'steps_per_sec': sps}
                row.update({f'info/{k}': float(v) for k, v in info.items()})
                jsonl.write(json.dumps(row) + '\n')
                jsonl.flush()
                if tb is not None:
                    for k, v in row.items():
                        if k.startswith('info/'):
                            tb.add_scalar(k, v, t + 1)
                    tb.flush()
            print('Step {}. Steps per sec: {:.4g}.'.format(t + 1, sps))
            if args.save_model:
                agent.save(os.path.join(log_path, 'agent.pt'))
    jsonl.close()
    if tb is not None:
        tb.close()
    print('Total time cost {:.4g}s.'.format(timer.time_cost()))
    return agent, evaluations


if __name__ == '__main__':
    run()
