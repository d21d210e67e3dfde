"""Python glue for the pt_hip C-ABI library (tests, bench and fixtures only; the product is the shared library)."""
from .scene import SceneData, load_scene_dir  # noqa: F401
