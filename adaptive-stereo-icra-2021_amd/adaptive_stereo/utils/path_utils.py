"""Repository-relative folders the reference's scripts ask for (adaptive_stereo/utils/path_utils.py:4-29):
``resources/`` for inputs shipped with the code, ``output/`` for what evaluate_model.py writes (:99)."""
import pathlib

_PACKAGE_PARENT = pathlib.Path(__file__).resolve().parents[2]     # the directory that holds ``adaptive_stereo``


def _under(name, reldir):
  return str(_PACKAGE_PARENT.joinpath(name, reldir) if reldir else _PACKAGE_PARENT / name)


def top_folder():
  return str(_PACKAGE_PARENT)


def resources_folder(reldir=""):
  return _under("resources", reldir)


def output_folder(reldir=""):
  return _under("output", reldir)
