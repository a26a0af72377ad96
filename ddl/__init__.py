"""Import alias for the framework package.

The source tree lives in ``distributed-deep-learning_amd/`` (the directory name
mandated for this project).  A hyphen is not a legal Python identifier, so this
tiny package re-points its ``__path__`` at that directory: ``import ddl.models``
loads ``distributed-deep-learning_amd/models``.
"""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "distributed-deep-learning_amd",
)
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
# one module identity for the directory: importing the hyphenated name through importlib yields THIS module
import sys as _sys
_sys.modules.setdefault("distributed-deep-learning_amd", _sys.modules[__name__])
