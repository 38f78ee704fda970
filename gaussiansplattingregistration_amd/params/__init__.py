from .merge_parameters import GaussianMixtureParams  # noqa: F401
from .registration_parameters import LocalRegistrationParams, MultiScaleRegistrationParams  # noqa: F401
