"""Scene file I/O of the reference's ``datasets`` package (inference side)."""
from .dataset_reader_physics import (Dataset, DatasetGroup, get_rollout, read_scene, write_results,  # noqa: F401
                                     write_results_npz, write_scene)
