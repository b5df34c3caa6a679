"""A `tensorflow` 1.x look-alike over float64 torch -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Purpose ("Oracle-B"): import the reference's learner -- agents/utils.py, agents/policies.py, agents/models.py and the
loop in utils.py -- UNMODIFIED and *execute* it, the same way oracle/fake_traci.py carries envs/*.py.  What the
reference itself wires on top of TensorFlow -- which slice of the observation feeds which tower
(agents/policies.py:99-118,191-211,227-235,345-353), the concat order of the hidden blocks, the gate order and the
done mask of the hand-unrolled LSTM (agents/utils.py:88-116), which state rows belong to pi / v and when they advance
(policies.py:125-136,153), the A2C and TD losses (policies.py:41-61,305-318), the reward normalisation / clipping
and the buffers (agents/models.py:174-229,319-376) -- is then pinned by the reference's own graph code.  Only the op
kernels below are restated, from TensorFlow 1.12's documented semantics:

  matmul, + - * /, relu / sigmoid / tanh / softmax, split / concat / squeeze / expand_dims / slicing, one_hot, log,
  clip_by_value, reduce_sum / mean / max, square, where, stop_gradient;
  tf.gradients ................. reverse-mode derivative (torch.autograd.grad of the same float64 graph);
  tf.clip_by_global_norm ....... norm = sqrt(sum ||t||^2); t * clip / max(norm, clip);
  tf.train.RMSPropOptimizer .... slots ms = 1, mom = 0;  ms += (1 - decay)(g^2 - ms);
                                 mom = momentum mom + lr g / sqrt(ms + epsilon);  var -= mom      (not centred);
  tf.train.AdamOptimizer ....... m, v = 0; lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); var -= lr_t m / (sqrt(v) + eps).

Lazy graph like TF1: ops build `Tensor` nodes (static shapes come from evaluating every op once on zeros, `None`
dimensions as 1), `Session.run(fetches, feed_dict)` evaluates them.  float32 placeholders round the fed value to
float32 first (TF casts the feed) and compute in float64 from there on.  `get_variable` calls a callable initializer
at creation time like TF1 does, so `np.random`-based initialisers (agents/utils.py:11-24) consume the global NumPy
stream in variable-creation order.

Install with `oracle.fake_tf.install()` (puts this module into sys.modules['tensorflow']).
"""
import contextlib
import sys
import types

import numpy as np
import torch

DT = torch.float64
__version__ = '1.12.0-fake'


class DType:
    def __init__(self, name, torch_dtype, np_dtype):
        self.name, self.torch, self.np = name, torch_dtype, np_dtype

    def __repr__(self):
        return 'tf.' + self.name


float32 = DType('float32', DT, np.float32)
float64 = DType('float64', DT, np.float64)
int32 = DType('int32', torch.long, np.int32)
int64 = DType('int64', torch.long, np.int64)
bool = DType('bool', torch.bool, np.bool_)          # noqa: A001  (tf.bool)


class Dimension:
    def __init__(self, value):
        self.value = value

    def __int__(self):
        return int(self.value)

    __index__ = __int__

    def __floordiv__(self, o):
        return Dimension(self.value // int(o))

    def __mul__(self, o):
        return Dimension(self.value * int(o))

    __rmul__ = __mul__

    def __add__(self, o):
        return Dimension(self.value + int(o))

    __radd__ = __add__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dimension) else o)

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return 'Dimension(%r)' % (self.value,)


class TensorShape:
    def __init__(self, dims):
        self.dims = [Dimension(d) for d in dims]

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)

    def as_list(self):
        return [d.value for d in self.dims]

    def __repr__(self):
        return 'TensorShape(%r)' % (self.as_list(),)


class Graph:
    def __init__(self):
        self.variables = {}          # full name -> Variable, creation order
        self.scope = []              # [(name, reuse)]
        self.seed = None
        self.grad_bundles = []       # every tf.gradients call (fixture recording reads the raw gradients)


_GRAPH = [Graph()]


def get_default_graph():
    return _GRAPH[0]


def reset_default_graph():
    _GRAPH[0] = Graph()


def set_random_seed(seed):
    _GRAPH[0].seed = seed


class ConfigProto:
    def __init__(self, **kw):
        self.kw = kw


# ---- nodes --------------------------------------------------------------------------------------
class Tensor:
    """A lazy node: value = fn(*input values)."""

    def __init__(self, fn, inputs=(), name=None, static=None):
        self.fn, self.inputs, self.name = fn, list(inputs), name
        self._static = static                       # example value (zeros) for static-shape queries
        if static is None:
            ex = [_example(i) for i in self.inputs]
            try:                                    # unknown static shape (None batch meeting a fixed one): like TF,
                with torch.no_grad():               # the shape is then only known at run time
                    self._static = None if any(e is None for e in ex) else fn(*ex)
            except (RuntimeError, IndexError):
                self._static = None

    @property
    def shape(self):
        ex = self._static
        if ex is None:
            raise ValueError('static shape unknown')
        dims = list(ex.shape) if torch.is_tensor(ex) else []
        none = getattr(self, '_none_dims', ())
        return TensorShape([None if i in none else d for i, d in enumerate(dims)])

    def get_shape(self):
        return self.shape

    # arithmetic
    def __add__(self, o):
        return _binary(torch.add, self, o)

    def __radd__(self, o):
        return _binary(torch.add, o, self)

    def __sub__(self, o):
        return _binary(torch.sub, self, o)

    def __rsub__(self, o):
        return _binary(torch.sub, o, self)

    def __mul__(self, o):
        return _binary(torch.mul, self, o)

    def __rmul__(self, o):
        return _binary(torch.mul, o, self)

    def __truediv__(self, o):
        return _binary(torch.div, self, o)

    def __rtruediv__(self, o):
        return _binary(torch.div, o, self)

    def __neg__(self):
        return Tensor(torch.neg, [self])

    def __getitem__(self, idx):
        return Tensor(lambda x: x[idx], [self])

    def __iter__(self):
        raise TypeError('Tensor objects are not iterable (as in TF1 graph mode)')

    __hash__ = object.__hash__


class Placeholder(Tensor):
    def __init__(self, dtype, shape=None, name=None):
        self.dtype = dtype
        shape = [] if shape is None else list(shape)
        self._none_dims = tuple(i for i, d in enumerate(shape) if d is None)
        ex = torch.zeros([2 if d is None else int(d) for d in shape], dtype=dtype.torch)   # unknown batch: any size > 1
        super().__init__(None, [], name, static=ex)
        self.decl_shape = shape

    def feed(self, value):
        a = np.asarray(value)
        if self.dtype in (float32, float64):
            a = np.asarray(a, self.dtype.np).astype(np.float64)        # TF casts the feed to the placeholder dtype
        elif self.dtype in (int32, int64):
            a = a.astype(np.int64)
        else:
            a = a.astype(np.bool_)
        t = torch.from_numpy(np.array(a, order='C'))           # (ascontiguousarray would promote 0-d to 1-d)
        want = self.decl_shape
        if len(want) != t.dim() or any(d is not None and int(d) != s for d, s in zip(want, t.shape)):
            raise ValueError('Cannot feed value of shape %r for Tensor %r, which has shape %r'
                             % (tuple(t.shape), self.name, tuple(want)))
        return t


class Variable(Tensor):
    def __init__(self, name, value, trainable=True):
        self.value = value                           # float64 torch tensor, the variable's current content
        self.var_name = name
        self.trainable = trainable
        super().__init__(None, [], name + ':0', static=value)

    @property
    def op(self):
        return types.SimpleNamespace(name=self.var_name)


class Operation(Tensor):
    """A node run for its side effect; evaluates to None."""

    def __init__(self, effect, inputs=()):
        self.effect = effect
        super().__init__(effect, inputs, static=torch.zeros(()))


def _example(x):
    return x._static if isinstance(x, Tensor) else x


def convert(x):
    if isinstance(x, Tensor):
        return x
    a = np.asarray(x)
    c = torch.from_numpy(np.array(a if a.dtype.kind == 'b' else a.astype(np.float64), order='C'))
    return Tensor(lambda: c, [], static=c)


def constant(value, dtype=None, shape=None, name=None):
    return convert(value)


def _binary(f, a, b):
    ins = [x for x in (a, b) if isinstance(x, Tensor)]
    if isinstance(a, Tensor) and isinstance(b, Tensor):
        return Tensor(lambda x, y: f(x, y), ins)
    if isinstance(a, Tensor):
        return Tensor(lambda x: f(x, torch.as_tensor(b, dtype=x.dtype) if not x.dtype.is_floating_point else b), ins)
    return Tensor(lambda y: f(torch.as_tensor(a, dtype=y.dtype if y.dtype.is_floating_point else DT), y), ins)


# ---- variables / scopes ---------------------------------------------------------------------------
class _Scope:
    def __init__(self, name, reuse):
        self.name, self.reuse = name, reuse


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    g = get_default_graph()
    parent_reuse = g.scope[-1].reuse if g.scope else None
    g.scope.append(_Scope(name, reuse if reuse is not None else parent_reuse))
    try:
        yield g.scope[-1]
    finally:
        g.scope.pop()


def constant_initializer(value=0.0):
    def init(shape, dtype=None, partition_info=None):
        return np.full(tuple(int(s) for s in shape), value, np.float32)
    return init


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    g = get_default_graph()
    full = '/'.join([s.name for s in g.scope] + [name])
    reuse = g.scope[-1].reuse if g.scope else None
    if full in g.variables:
        if not reuse:
            raise ValueError('Variable %s already exists, disallowed. Did you mean to set reuse=True?' % full)
        return g.variables[full]
    if reuse:
        raise ValueError('Variable %s does not exist, or was not created with tf.get_variable().' % full)
    dims = []
    for s in shape:
        s = s.value if isinstance(s, Dimension) else s
        if int(s) != s:
            raise TypeError('Dimension value must be integer or None, got %r' % (s,))
        dims.append(int(s))                            # tensor_shape.Dimension: int(value), must compare equal
    init = initializer if initializer is not None else constant_initializer(0.0)
    val = init(dims, dtype=float32, partition_info=None)     # TF1 calls a callable initializer at creation time
    val = torch.from_numpy(np.asarray(val, np.float32).astype(np.float64).reshape(dims).copy())   # float32 variable
    v = Variable(full, val, trainable)
    g.variables[full] = v
    return v


def trainable_variables(scope=None):
    """tf.trainable_variables(scope): collection filtered with re.match(scope, name) -- a prefix match."""
    import re
    out = []
    for n, v in get_default_graph().variables.items():
        if v.trainable and (scope is None or re.match(scope, n)):
            out.append(v)
    return out


def global_variables():
    return list(get_default_graph().variables.values())


def global_variables_initializer():
    return Operation(lambda: None)


# ---- ops ------------------------------------------------------------------------------------------------
def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


def matmul(a, b, name=None):
    return Tensor(lambda x, y: x @ y, [convert(a), convert(b)])


def tanh(x, name=None):
    return Tensor(torch.tanh, [convert(x)])


def log(x, name=None):
    return Tensor(torch.log, [convert(x)])


def exp(x, name=None):
    return Tensor(torch.exp, [convert(x)])


def square(x, name=None):
    return Tensor(lambda v: v * v, [convert(x)])


def sqrt(x, name=None):
    return Tensor(torch.sqrt, [convert(x)])


def clip_by_value(t, lo, hi, name=None):
    return Tensor(lambda v: torch.clamp(v, lo, hi), [convert(t)])


def stop_gradient(x, name=None):
    return Tensor(lambda v: v.detach(), [convert(x)])


def _axis(axis, kw):
    if axis is None:
        axis = kw.get('reduction_indices', kw.get('axis'))
    return axis


def reduce_sum(x, axis=None, keepdims=False, **kw):
    axis = _axis(axis, kw)
    return Tensor(lambda v: v.sum() if axis is None else v.sum(axis, keepdim=keepdims), [convert(x)])


def reduce_mean(x, axis=None, keepdims=False, **kw):
    axis = _axis(axis, kw)
    return Tensor(lambda v: v.mean() if axis is None else v.mean(axis, keepdim=keepdims), [convert(x)])


def reduce_max(x, axis=None, keepdims=False, **kw):
    axis = _axis(axis, kw)
    return Tensor(lambda v: v.max() if axis is None else v.max(axis, keepdim=keepdims).values, [convert(x)])


def one_hot(indices, depth, dtype=None, name=None):
    depth = int(depth)
    return Tensor(lambda i: torch.nn.functional.one_hot(i.long(), depth).to(DT), [convert(indices)])


def where(cond, x, y, name=None):
    return Tensor(lambda c, a, b: torch.where(c, a, b), [convert(cond), convert(x), convert(y)])


def squeeze(x, axis=None, name=None):
    return Tensor(lambda v: v.squeeze() if axis is None else v.squeeze(axis), [convert(x)])


def expand_dims(x, axis, name=None):
    return Tensor(lambda v: v.unsqueeze(axis), [convert(x)])


def reshape(x, shape, name=None):
    shape = [int(s) for s in shape]
    return Tensor(lambda v: v.reshape(shape), [convert(x)])


def concat(values=None, axis=None, name=None):
    vals = [convert(v) for v in values]
    return Tensor(lambda *vs: torch.cat(vs, int(axis)), vals)


def split(value=None, num_or_size_splits=None, axis=0, num=None, name=None):
    """Returns a Python list of tensors, like TF."""
    value = convert(value)
    size = value._static.shape[axis]
    if isinstance(num_or_size_splits, (int, np.integer, Dimension)):
        n = int(num_or_size_splits)
        if size % n:
            raise ValueError('Dimension size must be evenly divisible by %d but is %d' % (n, size))
        sizes = [size // n] * n
    else:
        sizes = [int(s) for s in num_or_size_splits]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    return [Tensor((lambda lo, hi: (lambda v: v.narrow(axis, lo, hi - lo)))(int(offs[k]), int(offs[k + 1])), [value])
            for k in range(len(sizes))]


def cast(x, dtype, name=None):
    return Tensor(lambda v: v.to(dtype.torch), [convert(x)])


def group(*ops, **kw):
    return Operation(lambda *a: None, [o for o in ops if isinstance(o, Tensor)])


def no_op(name=None):
    return Operation(lambda: None)


class _GradBundle:
    """tf.gradients(ys, xs): one autograd call shared by the per-variable output nodes."""

    def __init__(self, y, xs):
        self.node = Tensor(self._run, [y] + list(xs), static=torch.zeros(()))
        self.xs = list(xs)
        get_default_graph().grad_bundles.append(self)

    @staticmethod
    def _run(y, *xs):
        gs = torch.autograd.grad(y, xs, allow_unused=True, retain_graph=True)
        return [torch.zeros_like(x) if g is None else g.detach() for g, x in zip(gs, xs)]


def gradients(ys, xs, name=None, **kw):
    y = ys[0] if isinstance(ys, (list, tuple)) else ys
    b = _GradBundle(y, xs)
    return [Tensor((lambda k: (lambda allg: allg[k]))(k), [b.node], static=_example(x)) for k, x in enumerate(xs)]


def global_norm(t_list, name=None):
    ts = [convert(t) for t in t_list]
    return Tensor(lambda *v: torch.sqrt(sum((x * x).sum() for x in v)), ts)


def clip_by_global_norm(t_list, clip_norm, use_norm=None, name=None):
    """clip_ops.clip_by_global_norm: scale = clip_norm * min(1 / norm, 1 / clip_norm)."""
    norm = use_norm if use_norm is not None else global_norm(t_list)
    cn = float(clip_norm)
    outs = [Tensor(lambda v, n: v * (cn * torch.minimum(1.0 / n, torch.tensor(1.0 / cn, dtype=DT))), [convert(t), norm])
            for t in t_list]
    return outs, norm


nn = types.SimpleNamespace(
    relu=lambda x, name=None: Tensor(torch.relu, [convert(x)]),
    sigmoid=lambda x, name=None: Tensor(torch.sigmoid, [convert(x)]),
    tanh=tanh,
    softmax=lambda x, axis=-1, name=None: Tensor(lambda v: torch.softmax(v, axis), [convert(x)]),
    conv1d=None, conv2d=None,
)
sigmoid = nn.sigmoid


# ---- optimizers -------------------------------------------------------------------------------------------
class _Optimizer:
    def __init__(self):
        self.slots = {}               # var -> dict

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gv = [(g, v) for g, v in grads_and_vars if g is not None]
        extra = self._hyper_tensors()
        n = len(gv)

        def effect(*vals):
            gs, hyper = vals[:n], vals[n:]
            self._prepare(*hyper)
            for (_, v), g in zip(gv, gs):
                self._apply(v, g.detach())
            self._finish()
        for _, v in gv:
            self._create_slots(v)
        return Operation(effect, [g for g, _ in gv] + extra)

    def _prepare(self, *hyper):
        pass

    def _finish(self):
        pass


class RMSPropOptimizer(_Optimizer):
    """training/rmsprop.py + kernels/training_ops.cc ApplyRMSProp (TF 1.12)."""

    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, use_locking=False, centered=False,
                 name='RMSProp'):
        super().__init__()
        assert not centered
        self.lr, self.decay, self.momentum, self.epsilon = learning_rate, decay, momentum, epsilon

    def _hyper_tensors(self):
        return [convert(self.lr)]

    def _create_slots(self, v):
        self.slots[v] = dict(rms=torch.ones_like(v.value), momentum=torch.zeros_like(v.value))   # init_rms = ones

    def _prepare(self, lr):
        self._lr = float(lr)

    def _apply(self, v, g):
        s = self.slots[v]
        s['rms'] = s['rms'] + (g * g - s['rms']) * (1.0 - self.decay)
        s['momentum'] = s['momentum'] * self.momentum + self._lr * g / torch.sqrt(s['rms'] + self.epsilon)
        v.value = v.value - s['momentum']

    def get_slot(self, var, name):
        return self.slots[var][name]


class AdamOptimizer(_Optimizer):
    """training/adam.py + ApplyAdam (TF 1.12)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name='Adam'):
        super().__init__()
        self.lr, self.b1, self.b2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self.b1_power, self.b2_power = beta1, beta2

    def _hyper_tensors(self):
        return [convert(self.lr)]

    def _create_slots(self, v):
        self.slots[v] = dict(m=torch.zeros_like(v.value), v=torch.zeros_like(v.value))

    def _prepare(self, lr):
        self._lr_t = float(lr) * np.sqrt(1.0 - self.b2_power) / (1.0 - self.b1_power)

    def _apply(self, v, g):
        s = self.slots[v]
        s['m'] = s['m'] + (g - s['m']) * (1.0 - self.b1)
        s['v'] = s['v'] + (g * g - s['v']) * (1.0 - self.b2)
        v.value = v.value - self._lr_t * s['m'] / (torch.sqrt(s['v']) + self.epsilon)

    def _finish(self):
        self.b1_power *= self.b1
        self.b2_power *= self.b2

    def get_slot(self, var, name):
        return self.slots[var][name]


class Saver:
    """Only what agents/models.py:32,83-108 touches: save / restore all variables of the graph (an .npz)."""

    def __init__(self, var_list=None, max_to_keep=5):
        self.max_to_keep = max_to_keep

    def save(self, sess, save_path, global_step=None):
        path = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
        np.savez(path + '.npz', **{n.replace('/', '|'): v.value.numpy() for n, v in get_default_graph().variables.items()})
        return path

    def restore(self, sess, save_path):
        z = np.load(save_path + '.npz')
        for n, v in get_default_graph().variables.items():
            v.value = torch.from_numpy(z[n.replace('/', '|')].copy())


class Coordinator:
    def should_stop(self):
        return False


train = types.SimpleNamespace(RMSPropOptimizer=RMSPropOptimizer, AdamOptimizer=AdamOptimizer, Saver=Saver,
                              Coordinator=Coordinator)


class _FileWriter:
    def __init__(self, logdir=None, graph=None):
        self.logdir, self.records = logdir, []

    def add_summary(self, summ, global_step=None):
        self.records.append((global_step, summ))

    def flush(self):
        pass

    def close(self):
        pass


def _summary_scalar(name, tensor):
    t = convert(tensor)
    return Tensor(lambda v: (name, float(v.detach())), [t], static=torch.zeros(()))


def _summary_merge(inputs, name=None):
    return Tensor(lambda *v: list(v), list(inputs), static=torch.zeros(()))


summary = types.SimpleNamespace(scalar=_summary_scalar, merge=_summary_merge, FileWriter=_FileWriter,
                                merge_all=lambda: no_op())


# ---- session ------------------------------------------------------------------------------------------------
class Session:
    def __init__(self, target='', graph=None, config=None):
        self.config = config
        self.n_run = 0
        self.trace = None            # a list -> run() appends the float64 values of its fetches (fixture recording)
        self.last_memo = None

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        self.n_run += 1
        memo = {}
        for ph, val in (feed_dict or {}).items():
            if not isinstance(ph, Placeholder):
                raise TypeError('feed_dict keys must be placeholders')
            memo[id(ph)] = ph.feed(val)
        flat = []
        _flatten(fetches, flat)
        with torch.enable_grad():
            for f in flat:
                _evaluate(f, memo)
        self.last_memo = memo
        if self.trace is not None:
            self.trace.append([memo[id(f)].detach().numpy().copy() if torch.is_tensor(memo[id(f)]) else memo[id(f)] for f in flat])
        return _rebuild(fetches, memo)


def _flatten(f, out):
    if isinstance(f, (list, tuple)):
        for x in f:
            _flatten(x, out)
    elif isinstance(f, Tensor):
        out.append(f)
    else:
        raise TypeError('Fetch argument %r has invalid type' % (f,))


def _rebuild(f, memo):
    if isinstance(f, (list, tuple)):
        return [_rebuild(x, memo) for x in f]
    v = memo[id(f)]
    if isinstance(f, Operation):
        return None
    if torch.is_tensor(v):
        a = v.detach().numpy()
        if a.dtype == np.float64:
            a = a.astype(np.float32)                   # the graph is float32 in TF: fetched values are float32
        return a.copy() if a.ndim else a[()]
    return v


def _evaluate(root, memo):
    """Iterative post-order evaluation (the 120-step unrolled LSTM is deeper than Python's recursion limit)."""
    stack = [(root, False)]
    while stack:
        node, ready = stack.pop()
        k = id(node)
        if k in memo:
            continue
        if isinstance(node, Placeholder):
            raise ValueError('You must feed a value for placeholder tensor %r' % (node.name,))
        if isinstance(node, Variable):
            memo[k] = node.value.detach().clone().requires_grad_(True)
            continue
        if ready:
            memo[k] = node.fn(*[memo[id(i)] for i in node.inputs])
            continue
        stack.append((node, True))
        for i in node.inputs:
            if id(i) not in memo:
                stack.append((i, False))


def run64(fetches, feed_dict=None):
    """Session.run without the float32 rounding of the fetched values (fixture recording): float64 ndarrays."""
    memo = {id(ph): ph.feed(val) for ph, val in (feed_dict or {}).items()}
    flat = []
    _flatten(fetches, flat)
    with torch.enable_grad():
        for f in flat:
            _evaluate(f, memo)

    def get(f):
        if isinstance(f, (list, tuple)):
            return [get(x) for x in f]
        v = memo[id(f)]
        return v.detach().numpy().copy() if torch.is_tensor(v) else v
    return get(fetches)


def install():
    """Make `import tensorflow as tf` resolve to this module."""
    me = sys.modules[__name__]
    sys.modules['tensorflow'] = me
    return me
