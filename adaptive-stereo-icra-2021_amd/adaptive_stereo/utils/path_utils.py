"""Repository-relative folders the reference's scripts ask for (adaptive_stereo/utils/path_utils.py:4-29):
``resources/`` for inputs shipped with the code, ``output/`` for what evaluate_model.py writes (:99)."""
import os


def top_folder():
  """The directory that holds the ``adaptive_stereo`` package."""
  return os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def resources_folder(reldir=""):
  return os.path.join(top_folder(), "resources", reldir)


def output_folder(reldir=""):
  return os.path.join(top_folder(), "output", reldir)
