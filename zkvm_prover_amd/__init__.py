"""Import shim: the package directory is named ``zkvm-prover_amd`` (not a valid Python
identifier), so ``import zkvm_prover_amd`` forwards to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "zkvm-prover_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
