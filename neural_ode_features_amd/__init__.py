"""Importable alias of the package directory `neural-ode-features_amd/` (a hyphen
cannot appear in a Python module name).  All code lives there; this shim only
points the package path at it and runs its __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'neural-ode-features_amd')
__path__[:] = [_real]
with open(_os.path.join(_real, '__init__.py')) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _fh
