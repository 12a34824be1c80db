"""Importable alias of the ``dawn-pytorch_amd/`` source directory (a hyphen cannot appear in a Python
package name).  All code lives in ``dawn-pytorch_amd/``; this stub only redirects the package path."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "dawn-pytorch_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
