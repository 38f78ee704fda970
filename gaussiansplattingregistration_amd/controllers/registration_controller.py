"""Headless ``RegistrationController`` (reference ``src/controllers/registration_controller.py:24-28,93-120,145-163``)."""
from __future__ import annotations

from ..workers.registrators import LocalRegistrator, MultiScaleRegistratorMixture, MultiScaleRegistratorVoxel


class RegistrationController:
    def __init__(self, data_repository, ui_repository):
        self.data_repository = data_repository
        self.ui_repository = ui_repository
        self.errors = []

    def execute_local_registration_normal(self, params):
        repo = self.data_repository
        pc1 = repo.pc_open3d_list_first[repo.current_index]
        pc2 = repo.pc_open3d_list_second[repo.current_index]
        worker = LocalRegistrator(pc1, pc2, self.ui_repository.transformation_matrix, params)
        result = worker.run()
        self.handle_registration_result_local(result)
        return result

    def execute_multiscale_registration(self, use_corresponding, sparse_first, sparse_second, registration_type,
                                        relative_fitness, relative_rmse, voxel_values, iter_values, rejection_type, k_value,
                                        use_mixture=True):
        repo = self.data_repository
        if use_mixture:
            worker = MultiScaleRegistratorMixture(repo.pc_open3d_list_first, repo.pc_open3d_list_second,
                                                  self.ui_repository.transformation_matrix, use_corresponding, sparse_first,
                                                  sparse_second, registration_type, relative_fitness, relative_rmse,
                                                  voxel_values, iter_values, rejection_type, k_value)
        else:
            # the original clouds (the reference indexes the second list with [1], registration_controller.py:108 --
            # an IndexError without mixtures and the wrong cloud with them; the original cloud [0] is what is meant)
            worker = MultiScaleRegistratorVoxel(repo.pc_open3d_list_first[0], repo.pc_open3d_list_second[0],
                                                self.ui_repository.transformation_matrix, use_corresponding, sparse_first,
                                                sparse_second, registration_type, relative_fitness, relative_rmse,
                                                voxel_values, iter_values, rejection_type, k_value)
        result = worker.run()
        self.errors = worker.errors
        if result is not None:
            self.handle_registration_result_local(result)
        return result

    def handle_registration_result_local(self, result_data):       # :145-163
        self.ui_repository.transformation_matrix = result_data.result.transformation
