"""Deterministic "trained-like" weights and synthetic inputs.

The reference ships no checkpoint (reference README.md:10) and its full
state_dict is 49.4 M fp32 values, too large to commit.  Parity tests, the smoke
test and the benchmark therefore regenerate weights from a repo-owned,
counter-based generator that depends only on integer arithmetic plus float64
``log/sqrt/cos`` — the same values in the build container (where they are
loaded into the imported reference to make the golden fixtures) and on the GPU
box.  ``tests/golden/weights_checksums.json`` pins a per-tensor checksum.

Recipe: ``weight_v`` / plain ``weight`` ~ N(0, gain^2/fan_in)
with fan_in = numel(w[0]), gain 0.5 for enc_p / flow / enc_q and 1.0 for dec;
``bias`` ~ N(0, 0.05^2); ``weight_g`` = ||v|| per dim-0 slice times a keyed
factor in [0.8, 1.2] so that the weight-norm fold (w = g * v / ||v||) is
actually exercised (a factor of exactly 1 would hide a wrong norm axis on
ConvTranspose1d, whose dim 0 is the *input* channel).  Default init is useless
for parity: under weight-norm the reference's ``init_weights`` is a no-op and
outputs have rms ~2e-3.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _fnv1a64(s):
    h = 0xCBF29CE484222325
    for ch in s.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _key(seed, name):
    k = (_fnv1a64(name) ^ ((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF))
    return np.uint64(k & 0xFFFFFFFFFFFFFFFF)


def uniform01(seed, name, n):
    """n float64 uniforms in (0,1), keyed by (seed, name)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + _splitmix64(np.array([_key(seed, name)], dtype=np.uint64))[0]
    bits = _splitmix64(ctr) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed, name, shape):
    """Standard normals (Box-Muller in float64), float64 array of `shape`."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, 2 * n)
    z = np.sqrt(-2.0 * np.log(u[:n])) * np.cos(2.0 * np.pi * u[n:])
    return z.reshape(shape)


def _gain_for(name):
    return 1.0 if name.startswith("dec.") else 0.5


def fill_state_dict(shapes, seed=1234, gain_override=None):
    """Return {name: float32 ndarray} for an ordered {name: shape} mapping.

    `shapes` is typically ``{k: tuple(v.shape) for k, v in model.state_dict().items()}``.
    weight_g tensors are derived from their sibling weight_v.
    """
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight_g":
            continue
        if leaf in ("weight", "weight_v"):
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            gain = gain_override if gain_override is not None else _gain_for(name)
            w = normal(seed, name, shape) * (gain / np.sqrt(fan_in))
            out[name] = w.astype(np.float32)
        elif leaf == "bias":
            out[name] = (normal(seed, name, shape) * 0.05).astype(np.float32)
        elif leaf == "gamma":
            out[name] = (1.0 + 0.1 * normal(seed, name, shape)).astype(np.float32)
        elif leaf == "beta":
            out[name] = (0.1 * normal(seed, name, shape)).astype(np.float32)
        else:
            out[name] = (0.1 * normal(seed, name, shape)).astype(np.float32)
    res = {}
    for name, shape in shapes.items():
        if name.rsplit(".", 1)[-1] == "weight_g":
            v = out[name[:-1] + "v"].astype(np.float64)
            nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1))
            fac = 0.8 + 0.4 * uniform01(seed, name, v.shape[0])
            res[name] = (nrm * fac).astype(np.float32).reshape(tuple(int(s) for s in shape))
        else:
            res[name] = out[name]
    return res


def synthetic_mel(seed, B, T, n_mel=80):
    """mel[B,80,T] ~ N(-4, 1.5^2) clamped to [ln 1e-5, 2.5] (range of reference mel_processing.py:19-25)."""
    m = -4.0 + 1.5 * normal(seed, "mel", (B, n_mel, T))
    return np.clip(m, np.log(1e-5), 2.5).astype(np.float32)


def synthetic_eps(seed, B, T, C=192):
    """The reparameterisation noise the reference draws with randn_like (models.py:336)."""
    return normal(seed, "eps", (B, C, T)).astype(np.float32)


def ragged_lengths(seed, B, T, lo=0.6):
    u = uniform01(seed, "lengths", B)
    ln = np.maximum(1, np.round(T * (lo + (1.0 - lo) * u))).astype(np.int64)
    ln[0] = T
    return ln


def checksum(arr):
    """Order-sensitive 64-bit checksum of a float32 array's bit pattern."""
    b = np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32).astype(np.uint64).ravel()
    with np.errstate(over="ignore"):
        idx = np.arange(b.size, dtype=np.uint64)
        mixed = _splitmix64(b ^ (idx * np.uint64(0x9E3779B97F4A7C15)))
        return int(np.bitwise_xor.reduce(mixed)) if b.size else 0
