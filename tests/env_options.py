"""TEST HARNESS (not product): kernel choices through AUVP_<NAME> in os.environ on LIVE contexts.

The library reads AUVP_<NAME> once per handle, at auvp_create (option table); callers steer a live handle with
auvp_set_option / Context.set_option.  Many tests and probe scripts predate that API and flip os.environ between calls
(monkeypatch.setenv).  install() wraps the CDLL that auv_sim_amd._lib.load() returns in a proxy that pushes the current
environment into the handle's options before every call that takes a handle -- in THIS process only, and only for processes
that call install() (tests/conftest.py, tests/experiments/soak_*.py, some tools/*.py).  The product package has no such hook:
what a caller loads is the plain CDLL.  (Until round 5 the proxy lived in auv_sim_amd/_lib.py behind an environment variable.)
"""
import ctypes as C
import os


def install():
    """idempotent: auv_sim_amd._lib.load() returns the proxied library from now on"""
    from auv_sim_amd import _lib
    if getattr(_lib, "_env_options_installed", False):
        return
    plain_load = _lib.load

    def load():
        L = plain_load()
        if not isinstance(L, EnvOptionsLib):
            L = EnvOptionsLib(L, _lib.OPTION_NAMES)
            _lib._lib = L
        return L

    _lib.load = load
    _lib._env_options_installed = True


class EnvOptionsLib:
    """see the module docstring"""

    def __init__(self, lib, option_names):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_names", tuple(option_names))
        object.__setattr__(self, "_fns", {})
        object.__setattr__(self, "_seen", {})

    def _sync(self, args):
        if not args or not isinstance(args[0], C.c_void_p) or not args[0].value:
            return
        seen = self._seen.setdefault(args[0].value, {})
        for name in self._names:
            v = os.environ.get("AUVP_" + name)
            if name in seen and seen[name] == v:
                continue
            first = name not in seen
            seen[name] = v
            if v is None:
                if not first:  # (first sight and absent: nothing to take back)
                    self._lib.auvp_unset_option(args[0], name.encode())
            else:
                try:
                    iv = int(v)
                except ValueError:
                    iv = 0
                self._lib.auvp_set_option(args[0], name.encode(), iv)

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if name == "auvp_destroy":
            # the cache is keyed by the handle's address, which the allocator may hand out again: forget it with the handle
            def destroy(h, _f=f, _seen=self._seen):
                _seen.pop(getattr(h, "value", h), None)
                return _f(h)
            return destroy
        if not name.startswith("auvp_") or name in ("auvp_create", "auvp_set_option", "auvp_unset_option"):
            return f
        w = self._fns.get(name)
        if w is None:
            w = self._fns[name] = _SyncedFn(f, self._sync)
        return w

    def __setattr__(self, name, value):
        setattr(self._lib, name, value)


class _SyncedFn:
    def __init__(self, f, sync):
        object.__setattr__(self, "_f", f)
        object.__setattr__(self, "_sync", sync)

    def __call__(self, *args):
        self._sync(args)
        return self._f(*args)

    def __getattr__(self, name):
        return getattr(self._f, name)

    def __setattr__(self, name, value):
        setattr(self._f, name, value)
