"""Import shim: the package directory is ``mpc-rs_amd/`` (not a valid Python identifier), so this module makes it
importable as ``mpc_rs_amd`` by pointing the package path at it and running its ``__init__``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mpc-rs_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
