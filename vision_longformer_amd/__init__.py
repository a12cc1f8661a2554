"""Importable alias of the package directory ``vision-longformer_amd/`` (a hyphen
cannot appear in a Python module name).  All code lives there; this file only
points the import system at it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "vision-longformer_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
