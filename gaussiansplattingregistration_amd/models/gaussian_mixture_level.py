"""``GaussianMixtureModel`` record, as the reference's ``src/models/gaussian_mixture_level.py:1-7``."""


class GaussianMixtureModel:
    def __init__(self, xyz, colors, opacities, covariance, features):
        self.xyz = xyz
        self.covariance = covariance
        self.colors = colors
        self.opacities = opacities
        self.features = features
