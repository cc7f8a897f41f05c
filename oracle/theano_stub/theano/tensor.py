"""theano.tensor names used by gccNMF/realtime/gccNMFProcessor.py:256-269, evaluated with NumPy (see _lazy.py)."""
import numpy as np

from ._lazy import Variable, Input, as_variable


def tensor3(name=None, dtype='float64'):
    return Input(name, dtype, 3)


def matrix(name=None, dtype='float64'):
    return Input(name, dtype, 2)


def dot(a, b):
    # theano.tensor.dot == numpy.dot semantics (last axis of a with second-to-last of b), result dtype = upcast
    return Variable(lambda x, y: np.dot(x, y), [as_variable(a), as_variable(b)])


def argmax(x, axis=None, keepdims=False):
    return Variable(lambda v: np.argmax(v, axis=axis).astype(np.int64), [as_variable(x)])


def switch(cond, ift, iff):
    return Variable(lambda c, a, b: np.where(c, a, b), [as_variable(cond), as_variable(ift), as_variable(iff)])


def exp(x):
    return Variable(np.exp, [as_variable(x)])


def _sum(v, axis, keepdims):
    if v.dtype in (np.float16, np.float32):                                # Sum.acc_dtype: float64 accumulator, input dtype out
        return np.sum(v, axis=axis, keepdims=keepdims, dtype=np.float64).astype(v.dtype)
    return np.sum(v, axis=axis, keepdims=keepdims)


def sum(x, axis=None, keepdims=False, **kwargs):        # noqa: A001  (the reference calls tensor.sum)
    return Variable(lambda v: _sum(np.asarray(v), axis, keepdims), [as_variable(x)])
