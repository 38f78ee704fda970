"""Result records; field names of the reference's ``src/models/registration_data.py`` that the hot path fills."""
from dataclasses import dataclass, field
from typing import Any, List


@dataclass
class LocalRegistrationData:
    registration_type: str = ""
    initial_transformation: Any = None
    relative_fitness: float = 0.0
    relative_rmse: float = 0.0
    result_fitness: float = 0.0
    result_inlier_rmse: float = 0.0
    result_transformation: Any = None
    max_correspondence: float = 0.0
    max_iteration: int = 0


@dataclass
class MultiScaleRegistrationData:
    registration_type: str = ""
    initial_transformation: Any = None
    relative_fitness: float = 0.0
    relative_rmse: float = 0.0
    result_fitness: float = 0.0
    result_inlier_rmse: float = 0.0
    result_transformation: Any = None
    voxel_values: List[float] = field(default_factory=list)
    iteration_values: List[int] = field(default_factory=list)
    used_sparse_clouds: bool = False
    used_gaussian_mixtures: bool = True
