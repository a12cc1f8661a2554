"""Importable name of the package directory ``vision-longformer_amd/`` (a hyphen cannot appear in a Python module
name): sub-modules are found there.  Layout of that directory: its own ``__init__.py`` docstring and README.md."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vision-longformer_amd")]
__version__ = "0.2.0"
