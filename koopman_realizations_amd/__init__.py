"""Importable alias of the `koopman-realizations_amd/` package directory (a hyphen cannot
appear in a Python module name).  All code lives in ../koopman-realizations_amd/."""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_os.path.join(_os.path.dirname(_here), "koopman-realizations_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
