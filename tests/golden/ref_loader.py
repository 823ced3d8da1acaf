"""Import harness for the upstream reference (test infrastructure, build container only).

`/root/reference` does not exist on the GPU box; this module is used ONLY by
`tests/golden/make_golden.py` (fixture generation) and by the optional
`tests/test_oracle_vs_reference.py` (skipped when the reference is absent).

The reference needs `gym`, `rvo2` and `socialforce`, none of which are installed here and none of
which are on the hot path.  They are replaced by in-memory stub modules *before* the import
(SURVEY.md §8c).  Nothing under /root/reference is modified or copied.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("RGL_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "crowd_nav"))


def _install_stubs():
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env(object):
            pass

        registry = {}

        def register(id, entry_point, **kw):
            registry[id] = entry_point

        def make(id):
            mod_name, cls_name = registry[id].split(":")
            import importlib
            return getattr(importlib.import_module(mod_name), cls_name)()

        gym.Env = Env
        gym.make = make
        envs = types.ModuleType("gym.envs")
        registration = types.ModuleType("gym.envs.registration")
        registration.register = register
        envs.registration = registration
        gym.envs = envs
        sys.modules["gym"] = gym
        sys.modules["gym.envs"] = envs
        sys.modules["gym.envs.registration"] = registration
    for name in ("rvo2", "socialforce"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import matplotlib
    matplotlib.use("Agg")


def load_reference():
    """Make `crowd_nav` / `crowd_sim` importable from the read-only mount."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True          # never leave __pycache__ in the mount
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import crowd_nav.policy.policy_factory as pf  # noqa: F401  (registers all policies)
    return pf.policy_factory
