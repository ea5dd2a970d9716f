"""Import shim: makes the on-disk package directory ``localdiffusion-hallucination_amd/``
(the hyphen is part of the required repo layout and is not a legal Python identifier)
importable as ``localdiffusion_hallucination_amd``.

This module replaces itself in ``sys.modules`` with a real package object whose
``__path__`` is the hyphenated directory, so ``import localdiffusion_hallucination_amd.unet``
and relative imports inside the package work normally.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "localdiffusion-hallucination_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
