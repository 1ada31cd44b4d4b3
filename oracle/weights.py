"""TEST INFRASTRUCTURE ONLY.  Closed-form deterministic weights.

The same generator is evaluated on the reference side (when golden vectors are
made) and on the test side, so only outputs need to be stored in tests/golden/.
Values are computed in float64 and rounded once to float32.

  u(l, i) = splitmix64(l * 2**32 + i) >> 40, mapped to a uniform value in [-1, 1)   (integer hash:
             bit-reproducible on any machine, unlike sin() of large arguments)
  weight l, flat index i:  gain * sqrt(2 / fan_in) * sqrt(3) * u(l, i)      (unit-variance uniform x kaiming)
  bias   l, flat index i:  0.05 * u(l + 7919, i)

`l` is the position of the tensor in `named_parameters()` order among the
trainable (non-CEM-filter) parameters.  gain=0.1 reproduces the variance of the
reference's training init (kaiming fan_in x 0.1, networks.py:29-42,119); gain=1.0 gives
O(1) activations through the whole stack, which makes relative-error checks meaningful.
"""
import numpy as np
import torch


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _u(l, n):
    with np.errstate(over='ignore'):
        k = np.uint64(l) * np.uint64(1 << 32) + np.arange(n, dtype=np.uint64)
        h = _splitmix64(k) >> np.uint64(40)            # 24 random bits
    return h.astype(np.float64) / float(1 << 23) - 1.0   # [-1, 1)


def formula_tensor(shape, l, is_bias, gain=1.0):
    n = int(np.prod(shape))
    if is_bias:
        v = 0.05 * _u(l + 7919, n)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        v = gain * np.sqrt(2.0 / fan_in) * np.sqrt(3.0) * _u(l, n)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def fill_formula_weights(module, gain=1.0):
    """Overwrite every trainable parameter of `module` (CEM filter taps are frozen
    and skipped, as networks.py:29-31 skips them at init)."""
    l = 0
    with torch.no_grad():
        for name, p in module.named_parameters():
            if 'Filter_OP' in name or not p.requires_grad:
                continue
            p.copy_(formula_tensor(tuple(p.shape), l, is_bias=name.endswith('bias'), gain=gain))
            l += 1
    return l


def seeded_uniform(shape, seed, lo=0.0, hi=1.0):
    """Generator-independent seeded inputs (numpy PCG64, stable across versions)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((lo + (hi - lo) * rng.random(shape)).astype(np.float32))
