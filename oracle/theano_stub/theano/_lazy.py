"""TEST INFRASTRUCTURE (part of the oracle, never imported by the product): a NumPy-evaluated stand-in for the handful of
Theano names that ``gccNMF/realtime/gccNMFProcessor.py`` uses, so that the UNMODIFIED reference ``GCCNMFProcessor`` can be
imported and run in this container (Theano is not installable here) to generate golden vectors for the streaming path.

What the reference uses (gccNMFProcessor.py:139,195-198,202-203,222-224,239-270,273-276):
``theano.shared`` (+ ``get_value`` / ``set_value``, indexing, ``.conj()``, arithmetic with ndarrays, ``numpy.abs`` of a variable),
``theano.tensor.{tensor3, dot, argmax, switch, exp, sum}``, ``.T``, ``abs()``, ``<``, ``-  /  **  +``, ``theano.function`` and
``theano.compile.sharedvalue.SharedVariable``.  Expressions are recorded as a small graph and evaluated with NumPy when the
compiled function is called.  Type rules that differ between Theano and plain NumPy and that this stub mirrors:
  * Python scalars become typed constants the way ``theano.tensor.constant`` autocasts them with the default
    ``floatX = float64`` / ``cast_policy = custom``: the narrowest of int8/16/32/64 resp. float16/32/64 that holds the value
    exactly -- so ``switch(c, 1.0, 0.0)`` is float16 and ``1 + float32`` stays float32;
  * binary ops promote like ``theano.scalar.upcast`` == ``numpy.promote_types`` on the operand dtypes (a 0-d shared float32 is a
    float32 operand, not a weak scalar): ``int64 - float32`` is float64;
  * ``sum`` of float32 accumulates in float64 and returns float32 (``Sum.acc_dtype``);
  * ``SharedVariable.set_value`` converts to the variable's dtype only when that is lossless (``TensorType.filter``).
"""
import numpy as np


def _constant(x):
    """theano.tensor.constant autocasting (NumpyAutocaster, cast_policy='custom', floatX='float64')."""
    if isinstance(x, (bool, np.bool_)):
        return np.asarray(x, np.bool_)
    if isinstance(x, (int, np.integer)) and not isinstance(x, np.generic):
        for dt in (np.int8, np.int16, np.int32, np.int64):
            if np.asarray(x, dt) == x:
                return np.asarray(x, dt)
    if isinstance(x, float):
        for dt in (np.float16, np.float32, np.float64):
            v = np.asarray(x, dt)
            if v == x or (x != x):
                return v
    return np.asarray(x)


def as_variable(x):
    return x if isinstance(x, Variable) else Constant(_constant(x))


class Variable(object):
    """A node of the expression graph: ``fn(*evaluated_inputs)``."""
    __array_priority__ = 1000

    def __init__(self, fn=None, inputs=(), name=None):
        self.fn, self.inputs, self.name = fn, tuple(inputs), name

    def evaluate(self, env):
        if id(self) in env:
            return env[id(self)]
        v = self.fn(*[i.evaluate(env) for i in self.inputs])
        env[id(self)] = v
        return v

    # numpy ufuncs applied to a variable (numpy.abs(shared[0]), ndarray * variable) build graph nodes
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != '__call__' or kwargs:
            return NotImplemented
        return Variable(lambda *a: ufunc(*a), [as_variable(i) for i in inputs])

    def _bin(self, other, f, swap=False):
        a, b = as_variable(self), as_variable(other)
        return Variable(f, [b, a] if swap else [a, b])

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __pow__(self, o): return self._bin(o, np.power)
    def __rpow__(self, o): return self._bin(o, np.power, True)
    def __lt__(self, o): return self._bin(o, np.less)
    def __le__(self, o): return self._bin(o, np.less_equal)
    def __gt__(self, o): return self._bin(o, np.greater)
    def __ge__(self, o): return self._bin(o, np.greater_equal)
    def __neg__(self): return Variable(np.negative, [self])
    def __abs__(self): return Variable(np.abs, [self])
    def __getitem__(self, idx): return Variable(lambda a: a[idx], [self])
    def __iter__(self): raise TypeError('theano stub: variables are not iterable')
    def conj(self): return Variable(np.conj, [self])

    @property
    def T(self): return Variable(lambda a: a.T, [self])
    @property
    def real(self): return Variable(lambda a: a.real, [self])
    @property
    def imag(self): return Variable(lambda a: a.imag, [self])


class Constant(Variable):
    def __init__(self, value):
        Variable.__init__(self)
        self.value = value

    def evaluate(self, env):
        return self.value


class Input(Variable):
    """A free input of a compiled function (theano.tensor.tensor3 & co.)."""

    def __init__(self, name, dtype, ndim):
        Variable.__init__(self, name=name)
        self.dtype, self.ndim = np.dtype(dtype), ndim

    def evaluate(self, env):
        if id(self) not in env:
            raise ValueError('theano stub: missing input %r' % self.name)
        return env[id(self)]


class SharedVariable(Variable):
    def __init__(self, value, name=None):
        Variable.__init__(self, name=name)
        self.container = np.array(value, copy=True)

    def evaluate(self, env):
        return self.container

    def get_value(self, borrow=False):
        return self.container if borrow else self.container.copy()[()] if self.container.ndim == 0 else self.container.copy()

    def set_value(self, value, borrow=False):
        new = np.asarray(value)
        if new.ndim != self.container.ndim:
            raise TypeError('theano stub: wrong number of dimensions (%d, expected %d)' % (new.ndim, self.container.ndim))
        if new.dtype != self.container.dtype:
            conv = new.astype(self.container.dtype)
            if not np.array_equal(conv, new, equal_nan=True):           # TensorType.filter: only lossless conversions
                raise TypeError('theano stub: %s cannot be stored losslessly as %s' % (new.dtype, self.container.dtype))
            new = conv
        self.container = np.array(new, copy=True)


def shared(value, name=None, **kwargs):
    return SharedVariable(value, name)


class Function(object):
    def __init__(self, inputs, outputs):
        self.inputs = list(inputs)
        self.outputs, self.single = (list(outputs), False) if isinstance(outputs, (list, tuple)) else ([outputs], True)

    def __call__(self, *args):
        if len(args) != len(self.inputs):
            raise TypeError('theano stub: expected %d inputs, got %d' % (len(self.inputs), len(args)))
        env = {}
        for var, val in zip(self.inputs, args):
            val = np.asarray(val)
            if val.dtype != var.dtype:
                raise TypeError('theano stub: input %r expects %s, got %s' % (var.name, var.dtype, val.dtype))
            if val.ndim != var.ndim:
                raise TypeError('theano stub: input %r expects %d dimensions, got %d' % (var.name, var.ndim, val.ndim))
            env[id(var)] = val
        out = [np.asarray(as_variable(o).evaluate(env)) for o in self.outputs]
        return out[0] if self.single else out


def function(inputs=(), outputs=None, **kwargs):
    return Function(inputs, outputs)
