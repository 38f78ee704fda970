"""Headless ``GaussianMixtureWorker`` (reference ``src/gui/workers/downsampling/qt_gaussian_mixture.py:10-129``).

Same constructor arguments, same call order -- cloud 1 then cloud 2, each
``CreateMixtureLevel(xyz, DC colour, RAW opacity flattened, cov6, SH rest flattened)`` (``:42-47,66-71``) then
``CreateMixture(cluster_level, hem_reduction, distance_delta, color_delta, decay_rate, level)`` (``:55-58,79-82``),
then per level ``GaussianModel.from_mixture(level, sh_degree)`` and the point-cloud conversion (``:94-115``) --
but tensors stay on the GPU: no ``tolist()`` marshalling, no host round trip.  Cooperative cancel is
polled between stages as the reference does (``:36-39,50-53``).
"""
from __future__ import annotations

from .. import mixture_bind
from ..models.gaussian_mixture_level import GaussianMixtureModel
from ..models.gaussian_model import GaussianModel
from ..utils.point_cloud_converter import convert_gs_to_open3d_pc


class GaussianMixtureWorker:
    class ResultData:
        def __init__(self, list_gaussian_first, list_gaussian_second, list_open3d_first, list_open3d_second):
            self.list_gaussian_first = list_gaussian_first
            self.list_gaussian_second = list_gaussian_second
            self.list_open3d_first = list_open3d_first
            self.list_open3d_second = list_open3d_second

    def __init__(self, pc1, pc2, hem_reduction, distance_delta, color_delta, decay_rate, cluster_level,
                 progress=None, device_name=None):
        self.hem_reduction = hem_reduction
        self.distance_delta = distance_delta
        self.color_delta = color_delta
        self.decay_rate = decay_rate
        self.cluster_level = cluster_level
        self.gaussian_pc_first = pc1
        self.gaussian_pc_second = pc2
        self.current_progress = 0
        self.max_progress = 6
        self.signal_cancel = False
        self._progress = progress
        self.device_name = device_name or pc1.device_name
        self.stats = []

    def _mixture(self, pc):
        level = mixture_bind.MixtureLevel.CreateMixtureLevel(
            pc.get_xyz.detach(), pc.get_colors.detach(), pc.get_raw_opacity.detach().view(-1),
            pc.get_covariance(1).detach(), pc.get_spherical_harmonics.detach())
        self.update_progress()
        if self.signal_cancel:
            return None
        models = mixture_bind.MixtureCreator.CreateMixture(self.cluster_level, self.hem_reduction, self.distance_delta,
                                                           self.color_delta, self.decay_rate, level)
        self.stats.append(mixture_bind.MixtureCreator.last_stats)
        self.update_progress()
        return models

    def _convert(self, models, sh_degree):
        gaussians, clouds = [], []
        for mixture in models:
            mixture_model = GaussianMixtureModel(*mixture_bind.MixtureLevel.CreateArrays(mixture))
            gaussian = GaussianModel(device_name=self.device_name)
            gaussian.from_mixture(mixture_model, sh_degree)
            clouds.append(convert_gs_to_open3d_pc(gaussian))
            gaussians.append(gaussian)
        self.update_progress()
        return gaussians, clouds

    def run(self):
        if self.signal_cancel:
            return None
        first = self._mixture(self.gaussian_pc_first)
        if first is None or self.signal_cancel:
            return None
        second = self._mixture(self.gaussian_pc_second)
        if second is None or self.signal_cancel:
            return None
        sh_degree = self.gaussian_pc_first.sh_degree
        g1, o1 = self._convert(first, sh_degree)
        g2, o2 = self._convert(second, sh_degree)
        return GaussianMixtureWorker.ResultData(g1, g2, o1, o2)

    def update_progress(self):
        self.current_progress += 1
        if self._progress:
            self._progress(int(self.current_progress / self.max_progress * 100))

    def cancel(self):
        self.signal_cancel = True
