"""Headless ``DownsamplerController`` (reference ``src/controllers/downsampler_controller.py:20-41,74-87``)."""
from __future__ import annotations

from ..workers.gaussian_mixture import GaussianMixtureWorker


class DownsamplerController:
    def __init__(self, data_repository):
        self.data_repository = data_repository
        self.last_stats = None

    def create_mixture(self, params):
        repo = self.data_repository
        pc1 = repo.pc_gaussian_list_first[0]            # level 0 = original cloud (:23-26)
        pc2 = repo.pc_gaussian_list_second[0]
        worker = GaussianMixtureWorker(pc1, pc2, params.hem_reduction, params.distance_delta, params.color_delta,
                                       params.decay_rate, params.cluster_level)
        result = worker.run()
        self.last_stats = worker.stats
        if result is not None:
            self.handle_mixture_results(result)
        return result

    def handle_mixture_results(self, result_data):       # :74-87
        repo = self.data_repository
        repo.pc_gaussian_list_first = repo.pc_gaussian_list_first[:1]
        repo.pc_gaussian_list_second = repo.pc_gaussian_list_second[:1]
        repo.pc_open3d_list_first = repo.pc_open3d_list_first[:1]
        repo.pc_open3d_list_second = repo.pc_open3d_list_second[:1]
        repo.pc_gaussian_list_first.extend(result_data.list_gaussian_first)
        repo.pc_gaussian_list_second.extend(result_data.list_gaussian_second)
        repo.pc_open3d_list_first.extend(result_data.list_open3d_first)
        repo.pc_open3d_list_second.extend(result_data.list_open3d_second)
