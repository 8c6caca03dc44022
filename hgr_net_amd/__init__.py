"""Import alias for the ``hgr-net_amd/`` source directory.

The product package lives in ``hgr-net_amd/`` (a hyphen is not importable), so this
one-file package points its search path there; ``import hgr_net_amd.clip`` resolves to
``hgr-net_amd/clip/``.
"""
from pathlib import Path as _Path

_SRC = _Path(__file__).resolve().parent.parent / "hgr-net_amd"
__path__.insert(0, str(_SRC))
exec(compile((_SRC / "__init__.py").read_text(), str(_SRC / "__init__.py"), "exec"))
