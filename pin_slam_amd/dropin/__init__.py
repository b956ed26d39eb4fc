"""Drop-in mirrors of the reference's hot-path classes.

``install(reference_root)`` makes ``model.neural_points``, ``model.decoder``, ``utils.mapper``,
``utils.tracker`` and ``utils.mesher`` resolve to this package while every other ``model.*`` / ``utils.*`` /
``dataset.*`` module keeps resolving to the reference tree, so the reference's ``pin_slam.py``
runs unchanged.  Nothing is copied from the reference."""
from __future__ import annotations

import importlib
import os
import sys
import types

_OURS = ("model.decoder", "model.neural_points", "utils.tracker", "utils.mapper", "utils.mesher")


def install(reference_root: str):
    """Wire the module namespace for drop-in use.  Call before importing pin_slam."""
    reference_root = os.path.abspath(reference_root)
    if not os.path.isdir(os.path.join(reference_root, "utils")):
        raise FileNotFoundError(f"no PIN-SLAM tree at {reference_root}")
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    here = os.path.dirname(os.path.abspath(__file__))
    for pkg in ("model", "utils"):
        for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
            del sys.modules[k]
        m = types.ModuleType(pkg)
        # our directory first, the reference's second: same-named modules resolve to ours,
        # everything else (utils.config, utils.tools, utils.mesher, ...) to the reference
        m.__path__ = [os.path.join(here, pkg), os.path.join(reference_root, pkg)]
        m.__package__ = pkg
        sys.modules[pkg] = m
    out = {}
    for name in _OURS:  # order matters: utils.mapper looks for the reference Mapper to inherit
        real = importlib.import_module("pin_slam_amd.dropin." + name)
        sys.modules[name] = real
        setattr(sys.modules[name.split(".")[0]], name.split(".")[1], real)
        out[name] = real
    return out
