"""Drop-in for the reference's pybind11 module ``mixture_bind`` (``src/cpp_ext/mixture_bind.cpp:11-61``),
backed by the MI355X HEM kernels.

Same names, argument order and return shapes as the reference:

    level0 = MixtureLevel.CreateMixtureLevel(xyz, colors, opacities, covariance, features)
    levels = MixtureCreator.CreateMixture(clusterLevel, hemReduction, distanceDelta, colorDelta, decayRate, level0)
    xyz, colors, opacities, covariance, features = MixtureLevel.CreatePythonLists(levels[k])

``CreateMixture`` returns ``clusterLevel`` levels with level 0 removed (``mixture_wrapper.cpp:14-17``).
Arguments may be nested Python lists (as the reference requires), numpy arrays, or PyTorch tensors
-- device tensors are handed to the kernels without a copy.  ``MixtureLevel.CreateArrays(level)``
is the zero-marshalling sibling of ``CreatePythonLists``.

Wrong inner lengths raise ``RuntimeError`` with the reference's message (``include/vec.hpp:92-94,473-475``).

The reference draws parent flags from the process-global libc ``rand()`` stream, so a second cloud
continues where the first stopped (``qt_gaussian_mixture.py:55,79``).  This module keeps the same
process-wide stream position; ``reset_rng()`` rewinds it (a fresh reference process),
``set_rng_mode("hash")`` selects the counter-based device generator instead.
"""
from __future__ import annotations

import numpy as np

from . import hem as _hem

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_rng = {"mode": "glibc", "seed": 1, "position": 0}
_device = {"index": 0}


def reset_rng(seed: int = 1, position: int = 0):
    _rng["seed"], _rng["position"] = int(seed), int(position)


def set_rng_mode(mode: str):
    if mode not in ("glibc", "hash"):
        raise ValueError("mode must be 'glibc' or 'hash'")
    _rng["mode"] = mode


def rng_position() -> int:
    return _rng["position"]


def set_device(index: int):
    _device["index"] = int(index)


class vec3:
    def __init__(self, *a):
        if len(a) == 0:
            self.x = self.y = self.z = 0.0
        elif len(a) == 3:
            self.x, self.y, self.z = (float(v) for v in a)
        elif len(a) == 1:
            v = list(a[0])
            if len(v) != 3:
                raise RuntimeError("Python list must have exactly 3 elements.")
            self.x, self.y, self.z = (float(t) for t in v)
        else:
            raise TypeError("vec3(): incompatible constructor arguments")

    def __repr__(self):
        return "<vec3(%f, %f, %f)>" % (self.x, self.y, self.z)


class smat3:
    _names = ("e00", "e01", "e02", "e11", "e12", "e22")

    def __init__(self, *a):
        if len(a) == 0:
            vals = [0.0] * 6
        elif len(a) == 6:
            vals = [float(v) for v in a]
        elif len(a) == 1:
            vals = [float(v) for v in a[0]]
            if len(vals) != 6:
                raise RuntimeError("Python list must have exactly 6 elements.")
        else:
            raise TypeError("smat3(): incompatible constructor arguments")
        for n, v in zip(self._names, vals):
            setattr(self, n, v)

    def __repr__(self):
        return "<smat3(" + ", ".join("%f" % getattr(self, n) for n in self._names) + ")>"


class FeatureVector:
    def __init__(self, a=None):
        if a is None:
            self._v = []
        elif isinstance(a, int):
            self._v = [0.0] * a
        else:
            self._v = [float(v) for v in a]

    def GetSize(self):
        return len(self._v)

    def GetVector(self):
        return list(self._v)


def _is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _as_array(a, width, what):
    """nested list / ndarray / tensor -> (n, width) float32 array or tensor; reference error text on bad width."""
    if _is_tensor(a):
        t = a.detach()
        if width == 1:
            return t.reshape(-1).to(torch.float32)
        if t.dim() != 2 or (width > 0 and t.shape[1] != width):
            raise RuntimeError(f"Python list must have exactly {width} elements.")
        return t.to(torch.float32)
    if isinstance(a, np.ndarray):
        arr = a
    else:
        try:
            arr = np.asarray(a, dtype=np.float32)
        except ValueError:
            raise RuntimeError(f"Python list must have exactly {width} elements." if width > 1 else f"ragged {what}")
    if width == 1:
        return np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    if arr.size == 0:                     # empty level, or F = 0 feature rows: keep the row count
        rows = arr.shape[0] if arr.ndim >= 1 else 0
        return np.zeros((rows, arr.shape[1] if (arr.ndim == 2 and width == 0) else max(width, 0)), np.float32)
    if arr.ndim != 2 or (width > 0 and arr.shape[1] != width):
        raise RuntimeError(f"Python list must have exactly {width} elements.")
    return np.ascontiguousarray(arr, dtype=np.float32)


class MixtureLevel:
    """SoA level record; attribute names of ``hem::MixtureLevel`` (``mixturelevel.hpp:26-31``)."""

    def __init__(self):
        self.pointSet = np.zeros((0, 3), np.float32)
        self.colorSet = np.zeros((0, 3), np.float32)
        self.covarianceSet = np.zeros((0, 6), np.float32)
        self.opacities = np.zeros((0,), np.float32)
        self.features = np.zeros((0, 0), np.float32)

    def __len__(self):
        return int(self.pointSet.shape[0])

    @staticmethod
    def CreateMixtureLevel(xyz, colors, opacities, covariance, features):
        lv = MixtureLevel()
        lv.pointSet = _as_array(xyz, 3, "xyz")
        lv.colorSet = _as_array(colors, 3, "colors")
        lv.opacities = _as_array(opacities, 1, "opacities")
        lv.covarianceSet = _as_array(covariance, 6, "covariance")
        lv.features = _as_array(features, 0, "features")
        n = lv.pointSet.shape[0]
        for name in ("colorSet", "opacities", "covarianceSet", "features"):
            arr = getattr(lv, name)
            if arr.shape[0] != n and not (name == "features" and arr.shape[0] == 0 and n == 0):
                raise RuntimeError(f"MixtureLevel: {name} has {arr.shape[0]} entries, pointSet has {n}")
        return lv

    @staticmethod
    def CreateArrays(mixtureLevel):
        """(xyz, colors, opacities, covariance, features) as arrays/tensors, no list marshalling."""
        return (mixtureLevel.pointSet, mixtureLevel.colorSet, mixtureLevel.opacities, mixtureLevel.covarianceSet,
                mixtureLevel.features)

    @staticmethod
    def CreatePythonLists(mixtureLevel):
        def tl(a):
            if _is_tensor(a):
                return a.detach().cpu().tolist()
            return np.asarray(a).tolist()
        return (tl(mixtureLevel.pointSet), tl(mixtureLevel.colorSet), tl(mixtureLevel.opacities),
                tl(mixtureLevel.covarianceSet), tl(mixtureLevel.features))


class MixtureCreator:
    last_stats = None

    @staticmethod
    def CreateMixture(clusterLevel, hemReduction, distanceDelta, colorDelta, decayRate, mixtureLevel):
        on_dev = _is_tensor(mixtureLevel.pointSet) and mixtureLevel.pointSet.is_cuda
        device = mixtureLevel.pointSet.device.index if on_dev else _device["index"]
        out, stats = [], []
        with _hem.HemMixture(hemReduction, distanceDelta, colorDelta, decayRate, device=device, rng_mode=_rng["mode"],
                             rng_seed=_rng["seed"], rng_skip=_rng["position"]) as m:
            m.set_level0(mixtureLevel.pointSet, mixtureLevel.colorSet, mixtureLevel.opacities,
                         mixtureLevel.covarianceSet, mixtureLevel.features)
            for _ in range(int(clusterLevel)):
                m.run_level()
                st = m.stats()
                stats.append(st)
                d = m.get_level(as_torch=on_dev)
                lv = MixtureLevel()
                lv.pointSet, lv.colorSet, lv.opacities = d["xyz"], d["color"], d["opacity"]
                lv.covarianceSet, lv.features = d["cov6"], d["sh"]
                out.append(lv)
            _rng["position"] = m.stats()["rng_draws"]
        MixtureCreator.last_stats = stats
        return out
