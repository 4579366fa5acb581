"""Import shim: the package directory is named ``domain-rag_amd`` (not a valid Python
identifier), so ``import domain_rag_amd`` resolves here and extends ``__path__`` to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "domain-rag_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
