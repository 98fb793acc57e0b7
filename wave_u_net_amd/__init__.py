"""Importable alias of the `wave-u-net_amd/` package directory (a hyphen is not a valid
Python identifier).  All code lives in ../wave-u-net_amd/."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "wave-u-net_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
