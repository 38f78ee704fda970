"""Drop-in for the reference's ``src/utils/local_registration_util.py`` with the Open3D calls replaced
by the MI355X ICP kernels.

Same public names, argument meaning and error behaviour:

* ``KernelLossFunctionType`` / ``LocalRegistrationType`` -- integer ``.value`` = declaration order and
  ``.instance_name`` = the GUI label (reference ``:6-36``);
* ``get_rejection_loss`` (``:58-73``), ``get_estimation`` (``:39-51``), ``get_convergence_criteria`` (``:54-55``)
  return light descriptors instead of Open3D objects;
* ``do_icp_registration(point_cloud_first, point_cloud_second, init_transform, registration_params)``
  (``:76-100``) returns an object with ``.transformation`` (4x4 float64), ``.fitness``, ``.inlier_rmse``.
  The ten-positional-argument form the multiscale worker *intends*
  (``src/gui/workers/registration/qt_multiscale_registrator.py:84-87,136-138,218-220`` -- a ``TypeError``
  at the reference's HEAD) is accepted too.

Open3D raises ``RuntimeError`` for ``max_correspondence_distance <= 0`` and for point-to-plane ICP on
a target without normals; so does this module.  Generalized ICP (``registration_generalized_icp``, reference
``:96-98``) runs on the clouds' own covariances -- the reference's clouds always carry the splat covariances
(``point_cloud_converter.py:38``), which Open3D then uses as given; a cloud without covariances raises (Open3D
would estimate them from 20-nearest-neighbour normals, which this backend does not build).  Colored ICP
(``registration_colored_icp``, ``:92-94``) needs target normals and the colours of both clouds, as Open3D does.
"""
from __future__ import annotations

from enum import Enum

import numpy as np

from .. import icp as _icp


class KernelLossFunctionType(Enum):
    def __new__(cls, *args, **kwds):
        value = len(cls.__members__)
        obj = object.__new__(cls)
        obj._value_ = value
        return obj

    def __init__(self, name):
        self.instance_name = name

    Loss_None = "None"
    Tukey_Loss = "Tukey loss"
    Cauchy_Loss = "Cauchy loss"
    GMLoss = "GM loss"
    Huber_Loss = "Huber loss"


class LocalRegistrationType(Enum):
    def __new__(cls, *args, **kwds):
        value = len(cls.__members__)
        obj = object.__new__(cls)
        obj._value_ = value
        return obj

    def __init__(self, name):
        self.instance_name = name

    ICP_Point_To_Point = "Point-to-Point ICP"
    ICP_Point_To_Plane = "Point-to-Plane ICP"
    ICP_Color = "Colored ICP"
    ICP_General = "Generalized ICP"


class RobustLoss:
    """Stand-in for ``o3d.pipelines.registration.{L2,Tukey,Cauchy,GM,Huber}Loss``: (code, k)."""

    def __init__(self, code, k=0.0, name="L2Loss"):
        self.code, self.k, self.name = code, float(k), name

    def __repr__(self):
        return f"{self.name}(k={self.k})" if self.code else "L2Loss()"


class Estimation:
    """Stand-in for ``TransformationEstimationPointToPoint`` / ``PointToPlane(loss)``."""

    def __init__(self, kind, loss=None, name=""):
        self.kind, self.loss, self.name = kind, loss, name

    def __repr__(self):
        return f"{self.name}({self.loss!r})" if self.loss is not None else f"{self.name}()"


class ConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness, self.relative_rmse, self.max_iteration = relative_fitness, relative_rmse, max_iteration


def get_estimation(registration_type, loss_function):
    if loss_function is None:
        return Estimation(_icp.KIND_POINT_TO_POINT, None, "TransformationEstimationPointToPoint")
    if registration_type is LocalRegistrationType.ICP_Point_To_Point:
        return Estimation(_icp.KIND_POINT_TO_POINT, None, "TransformationEstimationPointToPoint")
    if registration_type is LocalRegistrationType.ICP_Point_To_Plane:
        return Estimation(_icp.KIND_POINT_TO_PLANE, loss_function, "TransformationEstimationPointToPlane")
    if registration_type is LocalRegistrationType.ICP_Color:
        return Estimation(_icp.KIND_COLORED, loss_function, "TransformationEstimationForColoredICP")
    if registration_type is LocalRegistrationType.ICP_General:
        return Estimation(_icp.KIND_GENERALIZED, loss_function, "TransformationEstimationForGeneralizedICP")
    return None


def get_convergence_criteria(relative_fitness, relative_rmse, max_iteration):
    return ConvergenceCriteria(relative_fitness, relative_rmse, max_iteration)


def get_rejection_loss(rejection_type, k_value, registration_type):
    if registration_type is LocalRegistrationType.ICP_Point_To_Point:
        return None
    if rejection_type is KernelLossFunctionType.Loss_None or k_value == 0.0:
        return RobustLoss(_icp.LOSS_L2, 0.0, "L2Loss")
    if rejection_type is KernelLossFunctionType.Tukey_Loss:
        return RobustLoss(_icp.LOSS_TUKEY, k_value, "TukeyLoss")
    if rejection_type is KernelLossFunctionType.Cauchy_Loss:
        return RobustLoss(_icp.LOSS_CAUCHY, k_value, "CauchyLoss")
    if rejection_type is KernelLossFunctionType.GMLoss:
        return RobustLoss(_icp.LOSS_GM, k_value, "GMLoss")
    if rejection_type is KernelLossFunctionType.Huber_Loss:
        return RobustLoss(_icp.LOSS_HUBER, k_value, "HuberLoss")
    return None


def registration_icp(source, target, max_correspondence_distance, init, estimation_method, criteria, device=None,
                     allreduce=None, n_source_global=None, ctx=None, allreduce_device=None, comm=None, target_prepared=False):
    """``o3d.pipelines.registration.registration_icp`` on ``PointCloud`` records (see ``point_cloud.py``).

    ``target_prepared``: the caller has already given THIS target (points, normals, ``max_correspondence_distance``) to ``ctx`` with
    ``ctx.set_target`` -- e.g. on a second stream while other work was running (bench.py builds every level's target index beside
    the HEM levels of the other cloud) -- so the index is not built again."""
    if not (max_correspondence_distance > 0.0):
        raise RuntimeError("[Open3D Error] Invalid max_correspondence_distance.")
    if estimation_method.kind in (_icp.KIND_POINT_TO_PLANE, _icp.KIND_COLORED) and not target.has_normals():
        raise RuntimeError("[Open3D Error] TransformationEstimationPointToPlane and "
                           "TransformationEstimationColoredICP require pre-computed normal vectors for target PointCloud.")
    if estimation_method.kind < 0:
        raise NotImplementedError(f"{estimation_method.name} is not part of this backend yet (SURVEY.md 8f, N2)")
    if estimation_method.kind == _icp.KIND_COLORED and (source.colors is None or target.colors is None):
        raise RuntimeError("[Open3D Error] ColoredICP requires color for both source and target PointCloud.")
    gicp_cov = {}
    if estimation_method.kind == _icp.KIND_GENERALIZED:
        # Open3D's InitializePointCloudForGeneralizedICP: pre-computed covariances are used as they are (every splat cloud,
        # point_cloud_converter.py:38); a cloud without them -- the reference's sparse input clouds, which reach this through
        # qt_multiscale_registrator.py:82-85 -- gets discs of thickness epsilon = 1e-3 perpendicular to its normals, and a cloud
        # without normals the 20-nearest-neighbour normals first
        for tag, pc in (("source", source), ("target", target)):
            if pc.has_covariances():
                gicp_cov[tag] = pc.cov6
            else:
                nrm = pc.normals if pc.has_normals() else _icp.normals_knn(pc.xyz32, knn=20, device=getattr(pc, "device_index", 0))
                gicp_cov[tag] = _icp.cov_from_normals(nrm, 1e-3, device=getattr(pc, "device_index", 0))
    if (comm is not None or allreduce_device is not None or allreduce is not None) and n_source_global is None:
        # fitness = correspondences / ALL source points: a shard cannot know the other shards' sizes
        raise RuntimeError("registration_icp: a sharded call (comm / allreduce) needs n_source_global, the source size over all ranks")
    dev = device if device is not None else getattr(target, "device_index", 0)
    loss = estimation_method.loss or RobustLoss(_icp.LOSS_L2)
    own = ctx is None            # a caller-provided context keeps its workspace across calls (no allocation in steady state)
    if own:
        ctx = _icp.IcpContext(device=dev)
    try:
        if (estimation_method.kind in (_icp.KIND_POINT_TO_POINT, _icp.KIND_POINT_TO_PLANE) and not target_prepared and comm is None
                and allreduce_device is None and allreduce is None and len(source) > 0
                and len({bool(getattr(a, "is_cuda", False)) for a in (source.xyz32, target.xyz32, target.normals) if a is not None}) == 1):
            # the plain single-process call: the two clouds go to the library in ONE call (gsr_icp_register_clouds), like Open3D's own
            # registration_icp(source, target, ...) -- no Python and no stream synchronisation between index build, source sort and loop
            r = ctx.register_clouds(source.xyz32, target.xyz32, target.normals if estimation_method.kind == _icp.KIND_POINT_TO_PLANE else None,
                                    max_correspondence_distance, np.asarray(init, dtype=np.float64), estimation_method.kind, loss.code, loss.k,
                                    criteria.relative_fitness, criteria.relative_rmse, criteria.max_iteration)
            res = _icp.RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"])
            res.timing = ctx.timing()
            return res
        if target_prepared:
            if ctx is None or own or getattr(ctx, "n_target", -1) != len(target):
                raise RuntimeError("registration_icp: target_prepared needs the caller's context with this target set")
        else:
            ctx.set_target(target.xyz32, target.normals if estimation_method.kind in (_icp.KIND_POINT_TO_PLANE, _icp.KIND_COLORED) else None,
                           max_correspondence_distance)
        ctx.set_source(source.xyz32)
        if estimation_method.kind == _icp.KIND_GENERALIZED:
            ctx.set_target_cov(gicp_cov["target"])
            ctx.set_source_cov(gicp_cov["source"])
        if estimation_method.kind == _icp.KIND_COLORED:
            ctx.set_target_color(target.colors)
            ctx.set_source_color(source.colors)
        # always (re)install the callbacks: a reused context must not keep a stale one from an earlier sharded call
        if comm is not None:
            ctx.set_comm(comm, n_source_global)
        elif allreduce_device is not None:
            ctx.set_comm(None, 0)
            ctx.set_allreduce_device(allreduce_device, n_source_global)
        else:
            ctx.set_comm(None, 0)
            ctx.set_allreduce_device(None, 0)
            ctx.set_allreduce(allreduce, n_source_global if allreduce is not None else 0)
        r = ctx.register(np.asarray(init, dtype=np.float64), estimation_method.kind, loss.code, loss.k,
                         criteria.relative_fitness, criteria.relative_rmse, criteria.max_iteration)
        res = _icp.RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"])
        res.timing = ctx.timing()
        return res
    finally:
        if own:
            ctx.close()


def registration_colored_icp(source, target, max_correspondence_distance, init, estimation_method, criteria, **kw):
    """``o3d.pipelines.registration.registration_colored_icp`` (Open3D ColoredICP.cpp: colour gradients of the target over
    Hybrid(2 * max_correspondence_distance, 30) neighbourhoods, then the ICP loop with a geometric and a photometric
    residual per pair, lambda_geometric = 0.968)."""
    if estimation_method.kind != _icp.KIND_COLORED:
        raise RuntimeError("registration_colored_icp needs TransformationEstimationForColoredICP")
    return registration_icp(source, target, max_correspondence_distance, init, estimation_method, criteria, **kw)


def registration_generalized_icp(source, target, max_correspondence_distance, init, estimation_method, criteria, **kw):
    """``o3d.pipelines.registration.registration_generalized_icp`` (Open3D GeneralizedICP.cpp: the ICP loop with
    ``TransformationEstimationForGeneralizedICP``) on clouds that carry covariances."""
    if estimation_method.kind != _icp.KIND_GENERALIZED:
        raise RuntimeError("registration_generalized_icp needs TransformationEstimationForGeneralizedICP")
    return registration_icp(source, target, max_correspondence_distance, init, estimation_method, criteria, **kw)


def do_icp_registration(point_cloud_first, point_cloud_second, init_transform, registration_params, *extra, **kw):
    """Reference signature (4 arguments) or the multiscale worker's intended 10-positional form:
    ``(pc1, pc2, T, registration_type, max_correspondence, relative_fitness, relative_rmse, max_iteration,
    rejection_type, k_value)``."""
    if extra:
        if len(extra) != 6:
            raise TypeError(f"do_icp_registration() takes 4 or 10 positional arguments but {4 + len(extra)} were given")
        from ..params.registration_parameters import LocalRegistrationParams
        registration_params = LocalRegistrationParams(registration_type=registration_params, max_correspondence=extra[0],
                                                      relative_fitness=extra[1], relative_rmse=extra[2], max_iteration=extra[3],
                                                      rejection_type=extra[4], k_value=extra[5])
    loss_function = get_rejection_loss(registration_params.rejection_type, registration_params.k_value,
                                       registration_params.registration_type)
    estimation_method = get_estimation(registration_params.registration_type, loss_function)
    convergence_criteria = get_convergence_criteria(registration_params.relative_fitness,
                                                    registration_params.relative_rmse,
                                                    registration_params.max_iteration)
    max_correspondence = registration_params.max_correspondence
    rt = registration_params.registration_type
    if rt in (LocalRegistrationType.ICP_Point_To_Point, LocalRegistrationType.ICP_Point_To_Plane):
        return registration_icp(point_cloud_first, point_cloud_second, max_correspondence, init_transform,
                                estimation_method, convergence_criteria, **kw)
    if rt is LocalRegistrationType.ICP_General:
        return registration_generalized_icp(point_cloud_first, point_cloud_second, max_correspondence, init_transform,
                                            estimation_method, convergence_criteria, **kw)
    if rt is LocalRegistrationType.ICP_Color:
        return registration_colored_icp(point_cloud_first, point_cloud_second, max_correspondence, init_transform,
                                        estimation_method, convergence_criteria, **kw)
    return None
