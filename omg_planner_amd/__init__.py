"""Import shim: the package directory is named ``omg-planner_amd`` (not a valid Python identifier);
``import omg_planner_amd`` resolves to it."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "omg-planner_amd"
__path__[:] = [str(_real)]
__file__ = str(_real / "__init__.py")
exec(compile((_real / "__init__.py").read_text(), __file__, "exec"))
