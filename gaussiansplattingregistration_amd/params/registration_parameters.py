"""ICP parameters: field names and defaults of the reference's ``src/params/registration_parameters.py:7-15``."""
from dataclasses import dataclass, field
from typing import List

from ..utils.local_registration_util import KernelLossFunctionType, LocalRegistrationType


@dataclass
class LocalRegistrationParams:
    registration_type: LocalRegistrationType = LocalRegistrationType.ICP_Point_To_Point
    max_correspondence: float = 5.0
    relative_fitness: float = 0.000001
    relative_rmse: float = 0.000001
    max_iteration: int = 30
    rejection_type: KernelLossFunctionType = KernelLossFunctionType.Loss_None
    k_value: float = 0.0


@dataclass
class MultiScaleRegistrationParams:
    """The argument list of ``signal_do_registration`` (``src/gui/tabs/multi_scale_registration_tab.py:13-15``)
    as one record; GUI defaults ``iter "50,30,20"`` / ``correspondences "5,2.5,2"`` (``:83,92``)."""
    use_corresponding: bool = False
    sparse_first: str = ""
    sparse_second: str = ""
    registration_type: LocalRegistrationType = LocalRegistrationType.ICP_Point_To_Point
    relative_fitness: float = 0.000001
    relative_rmse: float = 0.000001
    voxel_values: List[float] = field(default_factory=lambda: [5.0, 2.5, 2.0])
    iter_values: List[int] = field(default_factory=lambda: [50, 30, 20])
    rejection_type: KernelLossFunctionType = KernelLossFunctionType.Loss_None
    k_value: float = 0.0
    use_mixture: bool = True
