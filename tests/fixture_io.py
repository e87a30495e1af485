"""Load a golden fixture (tests/golden/*.npz) into oracle/HIP-ready inputs and expected outputs."""
import os, sys, json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
sys.path.insert(0, GOLD)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402


def cases():
    return sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith('.npz'))


def n_batches(alg, kw):
    if alg == 'sac':
        return 1
    if alg == 'spedersac':
        return 2 * (kw.get('extra_feature_steps', 1) + 1)
    return kw.get('extra_feature_steps', 1) + 1


def noise_plan(alg, S, A, B, kw):
    """Per train(): the ordered draws as ('idx', hi_key) / ('normal', shape) /
    ('randint', n)."""
    F = kw.get('feature_dim', 256)
    nf = kw.get('extra_feature_steps', 1) + 1
    plan = []
    if alg == 'sac':
        plan = [('idx',)]
    elif alg == 'vlsac':
        for _ in range(nf):
            plan += [('idx',), ('normal', (B, F))]
    elif alg == 'ctrlsac':
        plan = [('idx',)] * nf
    elif alg == 'spedersac':
        plan = [('idx',), ('idx',)] * nf
    elif alg == 'diffsrsac':
        for _ in range(nf):
            plan += [('idx',), ('randint', 1000), ('normal_scaled', (B, S))]
    plan += [('normal', (B, A)), ('normal', (B, A))]
    return plan


class Case:
    def __init__(self, name):
        from oracle.shapes import param_shapes
        self.name = name
        z = np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)
        self.z = z
        self.meta = json.loads(str(z['meta/json']))
        m = self.meta
        self.alg, self.S, self.A, self.B, self.T = m['alg'], m['S'], m['A'], m['B'], m['T']
        self.kw = dict(m['kwargs'])
        self.full = m['full']
        shape_kw = dict(self.kw)
        if m.get('patch_vae_hidden'):
            shape_kw['vae_hidden'] = m['patch_vae_hidden']
        self.shape_kw = shape_kw
        self.shapes = param_shapes(self.alg, self.S, self.A, **shape_kw)
        if self.full:
            self.replay = {k: z[f'replay/{k}'] for k in ('state', 'action', 'next_state', 'reward', 'done')}
            self.init = {k[5:]: z[k] for k in z.files if k.startswith('init/')}
        else:
            self.replay = synth.replay(self.S, self.A, m['replay_n'])
            self.init = synth.init_like(self.shapes)
            self._retie()
            self.init['log_alpha'] = np.log(np.float64(0.1))
            if self.alg == 'diffsrsac':
                self.init['noise_alphabars'] = z['init/noise_alphabars']
        # per-train inputs
        self.trains = []
        src = None if self.full else synth.NoiseSource()
        sigma = self.kw.get('sigma_scale_factor', 0.449)
        for t in range(self.T):
            idxs, eps = [], []
            if self.full:
                i = 0
                while f't{t}/idx/{i}' in z.files:
                    idxs.append(z[f't{t}/idx/{i}']); i += 1
                i = 0
                while f't{t}/eps/{i}' in z.files:
                    eps.append(z[f't{t}/eps/{i}']); i += 1
            else:
                for item in noise_plan(self.alg, self.S, self.A, self.B, self.kw):
                    if item[0] == 'idx':
                        idxs.append(src.indices(m['replay_n'], self.B))
                    elif item[0] == 'randint':
                        eps.append(src.indices(item[1], self.B))
                    elif item[0] == 'normal_scaled':
                        eps.append((np.float32(sigma) * src.normal(item[1])).astype(np.float32))
                    else:
                        eps.append(src.normal(item[1]))
            info = {k.split('/')[-1]: float(z[k]) for k in z.files if k.startswith(f't{t}/info/')}
            grads = {}
            for k in z.files:
                if k.startswith(f't{t}/grad/'):
                    _, _, opt, pname = k.split('/', 3)
                    grads.setdefault(opt, {})[pname] = z[k]
            self.trains.append(dict(idx=idxs, eps=eps, info=info, grads=grads))
        self.final = {k[6:]: z[k] for k in z.files if k.startswith('final/')}
        self.adam = {k[5:]: z[k] for k in z.files if k.startswith('adam/')}

    def _retie(self):
        """Mirror make_fixtures.load_synth_init: targets are hard copies of their sources."""
        P = self.init
        ties = [('critic', 'critic_target')]
        if self.alg == 'vlsac':
            ties.append(('f', 'f_target'))
            P['critic_target.noise'] = P['critic.noise'].copy()
        if self.alg in ('ctrlsac', 'spedersac'):
            ties.append(('phi', 'phi_target'))
        for s, d in ties:
            for k in list(P.keys()):
                if k.startswith(s + '.') and not k.endswith('noise'):
                    P[d + k[len(s):]] = P[k].copy()


def summary(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    head = np.zeros(16)
    head[:min(16, a.size)] = a[:16]
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], head])


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d = np.sqrt(((a - b) ** 2).sum())
    n = np.sqrt((b ** 2).sum())
    return d / n if n > 0 else d
