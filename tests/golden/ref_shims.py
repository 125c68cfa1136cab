"""Import shims that let the *pure-Python* layer of the reference run in this container.

Only used by the golden-vector generators in this directory (never by the product, never on
the GPU box).  The reference's native dependencies (mujoco, pinocchio, gym, gin, cv2, wandb)
are absent here; the shims below replace them with inert stubs so that the
reference's own controller / task-logic / metric code can be executed unmodified:

  * ``gin``: ``configurable`` injects the values parsed from
    controllers/Config/mujoco_controller_config.gin into the decorated constructors,
    ``parse_config_file`` is a no-op (MjFactory.py:24-28 does the real call).
  * ``np.NAN`` alias for NumPy 2 (Controller.py:16 uses it).
  * ``mujoco``, ``pinocchio``, ``gym``, ``cv2``, ``wandb``, ``glfw`` ...: stub packages created on
    demand by a meta-path finder; any attribute is an inert class.
"""
import ast
import importlib.abc
import importlib.machinery
import os
import re
import sys
import types

import numpy as np

REF = os.environ.get("D3IL_REFERENCE", "/root/reference")
GIN_FILE = os.path.join(REF, "environments/d3il/d3il_sim/controllers/Config/mujoco_controller_config.gin")


def _parse_gin(path):
    cfg = {}
    pat = re.compile(r"^\s*[\w\.]+\.(\w+)\.(\w+)\s*=\s*(.+)$")
    for line in open(path):
        if line.lstrip().startswith("#"):
            continue
        m = pat.match(line)
        if m:
            cfg.setdefault(m.group(1), {})[m.group(2)] = ast.literal_eval(m.group(3).strip())
    return cfg


class _AnyClass:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _AnyClass()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _AnyClass()


class _Anything(types.ModuleType):
    """Module stub: any attribute is an inert class (usable as a base class or callable)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        setattr(self, name, _AnyClass)
        return _AnyClass


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("mujoco", "pinocchio", "cv2", "wandb", "glfw", "gym", "OpenGL", "imageio", "open3d", "pybullet_utils", "pybullet")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Anything(spec.name)
        m.__path__ = []
        if spec.name == "gym":
            m.Env = type("Env", (), {})
        return m

    def exec_module(self, module):
        pass


_installed = {}


def install():
    if _installed:
        return _installed["cfg"]
    if not hasattr(np, "NAN"):
        np.NAN = np.nan
    d3il = os.path.join(REF, "environments", "d3il")
    for p in (d3il, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    cfg = _parse_gin(GIN_FILE)

    gin = types.ModuleType("gin")

    def configurable(cls):
        params = cfg.get(cls.__name__, {})
        orig = cls.__init__

        def __init__(self, *a, **k):
            kw = dict(params)
            kw.update(k)
            orig(self, *a, **kw)

        cls.__init__ = __init__
        return cls

    gin.configurable = configurable
    gin.parse_config_file = lambda *a, **k: None
    sys.modules["gin"] = gin
    sys.meta_path.append(_StubFinder())
    _installed["cfg"] = cfg
    return cfg
