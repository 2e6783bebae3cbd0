"""Import alias for the package directory ``ken-burns-effect_amd/``.

The repo layout contract names the package directory with a hyphen, which Python
cannot import directly.  Importing ``ken_burns_effect_amd`` executes this file,
which loads ``ken-burns-effect_amd/__init__.py`` as a regular package under the
importable name and replaces itself in ``sys.modules``; submodules then resolve
through the package's ``__path__`` as usual.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'ken-burns-effect_amd')
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, '__init__.py'),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
