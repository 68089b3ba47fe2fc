"""Deterministic, portable tensor recipe shared by the golden-vector generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Nothing under madeleine_amd/ imports this.

Golden fixtures under tests/golden/ hold *outputs* of the imported reference
(oracle/gen_golden.py).  The inputs and weights that produced them are regenerated on
both sides from this closed-form integer recipe (splitmix64 -> uniform floats), so the
fixtures stay KB-sized and do not depend on torch's RNG stream.
"""
import hashlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _key_seed(key: str) -> np.uint64:
    return np.uint64(int.from_bytes(hashlib.sha256(key.encode()).digest()[:8], "little"))


def uniform(shape, key: str, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 array of `shape`, i.i.d.-looking uniform in [lo, hi), a pure function of `key`."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + _key_seed(key))
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24-bit mantissa in [0,1)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def bernoulli(shape, key: str, p_one: float) -> np.ndarray:
    """float32 {0,1} array with P(1)=p_one."""
    return (uniform(shape, key, 0.0, 1.0) < p_one).astype(np.float32)


def state_dict_recipe(shapes: dict, tag: str = "w") -> dict:
    """Closed-form weights for every state_dict key.

    Linear weights/biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's default init scale;
    LayerNorm weight ~ 1 + U(-.1,.1), bias ~ U(-.1,.1); embedding ~ U(-1.7,1.7) (unit variance).
    `shapes` maps key -> shape (taken from a constructed module's state_dict()).
    """
    out = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        if k == "embedding.weight":
            out[k] = uniform(shp, f"{tag}:{k}", -1.7, 1.7)
        elif ".pre_attn.1." in k or ".pre_attn.5." in k or ".pre_attn.9." in k:  # LayerNorm
            if k.endswith("weight"):
                out[k] = 1.0 + uniform(shp, f"{tag}:{k}", -0.1, 0.1)
            else:
                out[k] = uniform(shp, f"{tag}:{k}", -0.1, 0.1)
        else:
            if k.endswith("weight"):
                fan_in = shp[-1]
            else:  # bias of a Linear: fan_in of the matching weight
                fan_in = shapes[k[: -len("bias")] + "weight"][-1]
            s = 1.0 / np.sqrt(float(fan_in))
            out[k] = uniform(shp, f"{tag}:{k}", -s, s)
    return out
