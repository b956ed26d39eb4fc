"""TEST INFRASTRUCTURE ONLY -- loads the *unmodified* PIN-SLAM reference modules on CPU.

Used only by ``oracle/make_golden.py`` (fixture generation) and by the
``-m "not gpu"`` tests that pin the numpy oracle against the real reference, and
only inside the build container: ``/root/reference`` does not exist on the GPU
box, so nothing that runs there may import this file (``available()`` says so).

The reference's hot-path modules (model/neural_points.py, model/decoder.py,
utils/tracker.py, utils/mapper.py) import a few packages that are not installed
here (open3d, roma, wandb, ...) at module top, none of which is touched inside
the hot-path functions.  We register permissive stub
modules for those names before importing.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("PIN_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "open3d", "open3d.visualization", "open3d.visualization.gui",
    "open3d.visualization.rendering", "open3d.geometry", "open3d.utility",
    "roma", "wandb", "natsort", "pyquaternion", "skimage", "skimage.measure",
    "laspy", "cv2", "gtsam", "evo", "pypose", "dtyper",
]


class _Anything(types.ModuleType):
    """Module whose every attribute is a harmless callable/class placeholder."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = _Placeholder(f"{self.__name__}.{name}")
        setattr(self, name, obj)
        return obj


class _Placeholder:
    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        return _Placeholder(self._name + "()")

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Placeholder(self._name + "." + name)

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):  # allows ``class X(stub.Base)``
        return (object,)


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


_loaded = {}


def load():
    """Import and return the reference modules as a namespace dict."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True  # never pollute the read-only tree
    for name in _STUBS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    # packages named model/ utils/ dataset/ must resolve to the reference
    for pkg in ("model", "utils", "dataset", "gui", "eval"):
        for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
            del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        cfg = importlib.import_module("utils.config")
        tools = importlib.import_module("utils.tools")
        loss = importlib.import_module("utils.loss")
        dec = importlib.import_module("model.decoder")
        npm = importlib.import_module("model.neural_points")
        trk = importlib.import_module("utils.tracker")
        mpr = importlib.import_module("utils.mapper")
    finally:
        sys.path.remove(REF_ROOT)
    _loaded.update(
        Config=cfg.Config, tools=tools, loss=loss, Decoder=dec.Decoder,
        NeuralPoints=npm.NeuralPoints, Tracker=trk.Tracker, tracker_mod=trk,
        Mapper=mpr.Mapper, mapper_mod=mpr,
    )
    return _loaded


class FakeDataset:
    """The attributes utils/mapper.py touches on its dataset (mapper.py:141-159, 212, 457-458)."""

    def __init__(self, n_frames=1):
        import numpy as np
        self.processed_frame = n_frames - 1
        self.odom_poses = np.tile(np.eye(4), (n_frames, 1, 1))
        self.pgo_poses = np.tile(np.eye(4), (n_frames, 1, 1))
        self.gt_poses = np.tile(np.eye(4), (n_frames, 1, 1))
        self.gt_pose_provided = True
        self.lose_track = False
        self.stop_status = False
        self.static_mask = None


def make_config(**over):
    """Reference Config on CPU with the overrides given as attributes."""
    import torch
    m = load()
    c = m["Config"]()
    c.device = "cpu"
    c.dtype = torch.float32
    c.silence = True
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(f"Config has no attribute {k}")
        setattr(c, k, v)
    # derived values (config.py:556-562)
    c.infer_bs = c.bs * 32
    return c
