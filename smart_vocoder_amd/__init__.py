"""Importable alias for the ``smart-vocoder_amd/`` package directory.

The product directory carries the reference's repo name (with a hyphen), which
Python cannot import directly; this alias package points its ``__path__`` at
that directory so ``import smart_vocoder_amd.models`` works.  The same
directory can also be put on ``sys.path`` directly so that the reference's
notebook imports (``import models, utils, commons`` — inference.ipynb cell 0)
resolve to this implementation.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_os.path.join(_os.path.dirname(_here), "smart-vocoder_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
