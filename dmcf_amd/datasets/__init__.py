"""Scene file I/O of the reference's ``datasets`` package (inference side)."""
from .dataset_reader_physics import (Dataset, get_rollout, read_scene, write_results, write_results_npz,  # noqa: F401
                                     write_scene)
