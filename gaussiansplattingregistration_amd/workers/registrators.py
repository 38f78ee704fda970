"""Headless ICP callers: ``LocalRegistrator`` (reference ``qt_local_registrator.py:26-32``) and
``MultiScaleRegistratorMixture`` (``qt_multiscale_registrator.py:163-246``), the coarse-to-fine loop over
HEM levels.  The multiscale worker reproduces the *intended* behaviour: the reference calls
``do_icp_registration`` with ten positional arguments (``:218-220``) although it takes four at HEAD.
Errors are returned as a list of strings where the reference emits ``signal_error``.
"""
from __future__ import annotations

import copy

from ..models.registration_data import LocalRegistrationData, MultiScaleRegistrationData
from ..utils.local_registration_util import do_icp_registration


class LocalRegistrator:
    class ResultData:
        def __init__(self, result, registration_data):
            self.result = result
            self.registration_data = registration_data

    def __init__(self, pc1, pc2, init_trans, registration_params):
        self.pc1, self.pc2, self.init_trans, self.params = pc1, pc2, init_trans, registration_params

    def run(self):
        p = self.params
        results = do_icp_registration(self.pc1, self.pc2, self.init_trans, p)
        data = LocalRegistrationData(registration_type=p.registration_type.instance_name,
                                     initial_transformation=self.init_trans, relative_fitness=p.relative_fitness,
                                     relative_rmse=p.relative_rmse, result_fitness=results.fitness,
                                     result_inlier_rmse=results.inlier_rmse, result_transformation=results.transformation,
                                     max_correspondence=p.max_correspondence, max_iteration=p.max_iteration)
        return LocalRegistrator.ResultData(results, data)


class MultiScaleRegistratorMixture:
    class ResultData:
        def __init__(self, result, registration_data):
            self.result = result
            self.registration_data = registration_data

    def __init__(self, pc1_list, pc2_list, init_trans, use_corresponding, sparse_first, sparse_second, registration_type,
                 relative_fitness, relative_rmse, voxel_values, iter_values, rejection_type, k_value, progress=None):
        self.pc1_list, self.pc2_list = pc1_list, pc2_list
        self.init_trans = init_trans
        self.use_corresponding = use_corresponding
        self.sparse_first_path, self.sparse_second_path = sparse_first, sparse_second
        self.registration_type = registration_type
        self.relative_fitness, self.relative_rmse = relative_fitness, relative_rmse
        self.voxel_values, self.iter_values = voxel_values, iter_values
        self.rejection_type, self.k_value = rejection_type, k_value
        self.errors = []
        self.signal_cancel = False
        self._progress = progress
        self.level_results = []

    def _check_valid_data(self):            # qt_multiscale_registrator.py:173-195
        if len(self.pc1_list) != len(self.pc2_list):
            self.errors.append("The two point cloud lists differ in size.")
            return False
        if len(self.pc1_list) <= 1:
            self.errors.append("There are no downscaled mixtures. First create Gaussian Mixtures by "
                               "running the HEM algorithm in the \"Mixture\" tab!")
            return False
        if len(self.iter_values) != len(self.voxel_values):
            self.errors.append("The number of iteration and voxel values provided do not match.")
            return False
        if len(self.iter_values) != len(self.pc1_list):
            self.errors.append("The number of iterations and the mixture levels do not match.")
            return False
        return True

    def _register_sparse_point_clouds(self):         # qt_multiscale_registrator.py:74-90
        """Pre-registration on the two sparse input clouds (``x y z red green blue`` .ply, e.g. COLMAP's) with the first
        correspondence distance and iteration count of the lists; its transformation seeds the multiscale loop."""
        from ..utils.file_loader import load_sparse_pc
        sparse_pc1 = load_sparse_pc(self.sparse_first_path)
        sparse_pc2 = load_sparse_pc(self.sparse_second_path)
        if not sparse_pc1 or not sparse_pc2:
            self.errors.append("Point clouds provided as sparse were of a different type")
            return None
        try:
            sparse_result = do_icp_registration(sparse_pc1, sparse_pc2, self.init_trans, self.registration_type,
                                                self.voxel_values[0], self.relative_fitness, self.relative_rmse,
                                                self.iter_values[0], self.rejection_type, self.k_value)
        except RuntimeError as e:
            self.errors.append(f"{e}\nSource: \"{sparse_pc1}\"\nTarget: \"{sparse_pc2}\"")
            return None
        self.sparse_result = sparse_result
        if self._progress:
            self._progress(int(1 / (len(self.iter_values) + 1) * 100))
        return sparse_result.transformation

    def _register_main_point_clouds(self, initial_transformation):   # :197-236
        current_trans = initial_transformation
        results = None
        for index in range(len(self.iter_values)):
            if self.signal_cancel:
                return None
            max_iter = self.iter_values[index]
            max_correspondence = self.voxel_values[index]
            pc1 = self.pc1_list[-(index + 1)]        # last list entry = coarsest level
            pc2 = self.pc2_list[-(index + 1)]
            try:
                results = do_icp_registration(pc1, pc2, current_trans, self.registration_type, max_correspondence,
                                              self.relative_fitness, self.relative_rmse, max_iter, self.rejection_type,
                                              self.k_value)
            except RuntimeError as e:
                self.errors.append(f"{e}\nSource: \"{pc1}\"\nTarget: \"{pc2}\"")
                return None
            self.level_results.append(results)
            if self._progress:
                self._progress(int((index + 1) / len(self.iter_values) * 100))
            current_trans = results.transformation
        return results

    def run(self):
        if self._check_valid_data() is False:
            return None
        current_trans = copy.deepcopy(self.init_trans)
        if self.use_corresponding:                      # qt_multiscale_registrator.py:45-46
            current_trans = self._register_sparse_point_clouds()
            if current_trans is None:
                return None
        if self.signal_cancel:
            return None
        results = self._register_main_point_clouds(current_trans)
        if results is None:
            return None
        data = MultiScaleRegistrationData(registration_type=self.registration_type.instance_name,
                                          initial_transformation=self.init_trans, relative_fitness=self.relative_fitness,
                                          relative_rmse=self.relative_rmse, result_fitness=results.fitness,
                                          result_inlier_rmse=results.inlier_rmse,
                                          result_transformation=results.transformation, voxel_values=self.voxel_values,
                                          iteration_values=self.iter_values, used_sparse_clouds=self.use_corresponding,
                                          used_gaussian_mixtures=True)
        return MultiScaleRegistratorMixture.ResultData(results, data)

    def cancel(self):
        self.signal_cancel = True


class MultiScaleRegistratorVoxel(MultiScaleRegistratorMixture):
    """Voxel multiscale registration (reference ``qt_multiscale_registrator.py:102-160``): per scale both clouds are
    voxel-down-sampled at ``voxel_values[scale]``, normals come from the averaged covariances (Open3D's
    ``estimate_normals`` uses a cloud's covariances when it has them, so the hybrid search parameter of the reference
    call is inert), and ICP runs with ``max_correspondence = voxel_values[scale]``."""

    def __init__(self, pc1, pc2, init_trans, use_corresponding, sparse_first, sparse_second, registration_type,
                 relative_fitness, relative_rmse, voxel_values, iter_values, rejection_type, k_value, progress=None):
        super().__init__([pc1], [pc2], init_trans, use_corresponding, sparse_first, sparse_second, registration_type,
                         relative_fitness, relative_rmse, voxel_values, iter_values, rejection_type, k_value, progress)
        self.pc1, self.pc2 = pc1, pc2

    def _check_valid_data(self):            # :116-122
        if len(self.iter_values) != len(self.voxel_values):
            self.errors.append("The number of iteration and voxel values provided do not match.")
            return False
        return True

    def _register_main_point_clouds(self, initial_transformation):   # :124-150
        current_trans = initial_transformation
        results = None
        for scale in range(len(self.iter_values)):
            if self.signal_cancel:
                return None
            max_iter, radius = self.iter_values[scale], self.voxel_values[scale]
            source_down = target_down = None
            try:
                source_down = self.pc1.voxel_down_sample(radius)
                target_down = self.pc2.voxel_down_sample(radius)
                source_down.estimate_normals()
                target_down.estimate_normals()
                results = do_icp_registration(source_down, target_down, current_trans, self.registration_type, radius,
                                              self.relative_fitness, self.relative_rmse, max_iter, self.rejection_type,
                                              self.k_value)
            except RuntimeError as e:
                self.errors.append(f"{e}\nSource: \"{source_down}\"\nTarget: \"{target_down}\"")
                return None
            self.level_results.append(results)
            if self._progress:
                self._progress(int((scale + 1) / len(self.iter_values) * 100))
            current_trans = results.transformation
        return results

    def run(self):
        out = super().run()
        if out is not None:
            out.registration_data.used_gaussian_mixtures = False
        return out
