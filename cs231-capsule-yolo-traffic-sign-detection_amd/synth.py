"""Synthetic GTSRB / GTSDB-shaped data (the reference ships none, .gitignore:1); shapes and dtypes of
what build_data.py emits.  Sample n is generated identically for every world size
so that data-parallel shards union to the single-process dataset."""
import numpy as np


def images(n, hw, seed=1234, first=0):
    """uint8 U{0..255} NHWC -> (x-128)/128 float32 (utils.py:122-123). Sample i depends only on (seed, first+i)."""
    out = np.empty((n, hw, hw, 3), dtype=np.float32)
    for i in range(n):
        rng = np.random.default_rng([seed, first + i])
        out[i] = (rng.integers(0, 256, (hw, hw, 3), dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    return out


def gtsrb_labels(n, n_classes=43, seed=1234, first=0):
    return np.array([np.random.default_rng([seed, 7, first + i]).integers(0, n_classes) for i in range(n)],
                    dtype=np.int64)


def gtsdb_labels(n, g, n_classes, seed=1234, first=0):
    """float64 [n,g,g,5+C]: 1..3 object cells per image: [1, xc, yc, w, h] + one-hot class (build_data.py:84-103)."""
    y = np.zeros((n, g, g, 5 + n_classes), dtype=np.float64)
    for i in range(n):
        rng = np.random.default_rng([seed, 11, first + i])
        k = min(int(rng.integers(1, 4)), g * g)
        for c in rng.choice(g * g, size=k, replace=False):
            r, col = divmod(int(c), g)
            y[i, r, col, 0] = 1.0
            y[i, r, col, 1:3] = rng.uniform(0.0, 1.0, 2)
            y[i, r, col, 3:5] = rng.uniform(0.02, 0.15, 2)
            if n_classes > 0:
                y[i, r, col, 5 + int(rng.integers(0, n_classes))] = 1.0
    return y
