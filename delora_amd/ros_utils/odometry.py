"""Scan-to-scan odometry on a stream of raw LiDAR scans: the ROS-free core of the reference's inference node.

``ScanToScanOdometry`` is what ``OdometryPublisher.subscriber_callback`` / ``predict_and_publish`` do between receiving
a PointCloud2 message and filling the Odometry message (reference src/ros_utils/odometry_publisher.py:93-172), and
``TrajectoryIntegrator`` is ``OdometryIntegrator.update_transformation`` (src/ros_utils/odometry_integrator.py:79-93);
the ROS plumbing around them (publishers, TF, message conversion) is out of scope and not built.  Both scans of a pair are projected by ONE launch of the
HIP projection (``geometry.project``), the network runs on the stacked pair, nothing else touches the points.

``quaternion_from_matrix`` / ``quaternion_matrix`` restate the two ``tf.transformations`` functions the node calls
(:151, odometry_integrator.py:82,86; ROS ``tf`` is third party and absent here: restated from the published algorithm of
transformations.py as shipped with ROS tf, quaternions in (x, y, z, w) order; checked against scipy's Rotation in
tests/test_host_logic.py -- parity with the ROS package itself is unpinned).
"""
import math

import numpy as np
import torch

from .. import geometry
from ..models import model as model_module
from ..models import model_parts


def quaternion_from_matrix(matrix):
    """(x, y, z, w) of the rotation part of a 4x4 homogeneous matrix (tf.transformations.quaternion_from_matrix)."""
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    q = np.empty((4,), dtype=np.float64)
    t = np.trace(M)
    if t > M[3, 3]:
        q[3] = t
        q[2] = M[1, 0] - M[0, 1]
        q[1] = M[0, 2] - M[2, 0]
        q[0] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    q *= 0.5 / math.sqrt(t * M[3, 3])
    return q


def quaternion_matrix(quaternion):
    """4x4 homogeneous rotation matrix of an (x, y, z, w) quaternion (tf.transformations.quaternion_matrix)."""
    q = np.array(quaternion[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= math.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array((
        (1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
        (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
        (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
        (0.0, 0.0, 0.0, 1.0)), dtype=np.float64)


class TrajectoryIntegrator(object):
    """Accumulated sensor pose ``T_0_t`` as the product of the scan-to-scan transforms (odometry_integrator.py:45,79-86)."""

    def __init__(self):
        self.T_0_t = np.expand_dims(np.eye(4), axis=0)

    def update_transformation(self, quaternion, translation):
        T_t_1_t = np.zeros((1, 4, 4))
        T_t_1_t[0] = quaternion_matrix(quaternion)
        T_t_1_t[0, :3, 3] = translation
        self.T_0_t = np.matmul(self.T_0_t, T_t_1_t)
        return self.T_0_t[0, :3, 3].copy(), quaternion_from_matrix(self.T_0_t[0])


def filter_scans(scan):
    """``[1,C,N]`` numpy scan without the points that have an exactly-zero x, y or z or lie within 0.3 m of the sensor
    (odometry_publisher.py:93-102)."""
    scan = np.asarray(scan)
    xyz = scan[0, :3]
    keep = (xyz[0] != 0.0) & (xyz[1] != 0.0) & (xyz[2] != 0.0)
    keep &= np.linalg.norm(xyz, axis=0) > 0.3
    return scan[:, :, keep]


def _project_pair_hip(previous, current, sensor):
    """Both scans through one launch of the HIP projection -> ``[2,4,H,W]`` (x, y, z, range)."""
    pts = torch.cat((previous[0, :3], current[0, :3]), dim=1).contiguous().float()
    n0, n1 = previous.shape[2], current.shape[2]
    offs = torch.tensor([0, n0, n0 + n1], dtype=torch.int32, device=pts.device)
    return geometry.project(pts, offs, max(n0, n1), sensor)["image4"]


class ScanToScanOdometry(object):
    """Feed raw scans with ``push``; from the second scan on every call returns the motion since the previous scan.

    config: the dict of ``bin/run_rosnode.py`` (the training config plus ``checkpoint``, ``datasets=[name]``,
    ``integrate_odometry``).  ``model`` / ``project_pair`` can be injected (tests run the core on the CPU with the
    oracle's projection); by default the model is built from the config and loaded from ``config["checkpoint"]`` and
    the projection is the HIP one."""

    def __init__(self, config, model=None, project_pair=None):
        self.config = config
        self.device = config["device"]
        self.dataset = config["datasets"][0]                       # "Assumes the dataset in config['datasets'][0]" (:26)
        self.sensor = geometry.Sensor.from_config(config, self.dataset)
        if model is None:
            model = model_module.OdometryModel(config=config).to(self.device)
            if config.get("checkpoint"):
                state = torch.load(config["checkpoint"], map_location=self.device, weights_only=False)
                model.load_state_dict(state["model_state_dict"])
        self.model = model.eval()
        self.geometry_handler = model_parts.GeometryHandler(config=config)
        self.project_pair = project_pair if project_pair is not None else _project_pair_hip
        self.integrator = TrajectoryIntegrator() if config.get("integrate_odometry", True) else None
        self.scaling_factor = 1.0
        self.point_cloud_t_1 = None

    def normalize_input(self, input_1, input_2):
        """Both clouds divided by the mean range over all their points, in place (odometry_publisher.py:104-113)."""
        mean_range = torch.mean(torch.cat((torch.norm(input_1, dim=1), torch.norm(input_2, dim=1)), dim=1))
        input_1 /= mean_range
        input_2 /= mean_range
        return float(mean_range)

    @torch.no_grad()
    def push(self, scan):
        """scan: ``[1,C>=3,N]`` (x, y, z first); a numpy array is filtered with ``filter_scans`` as the node does with every
        message, a tensor is taken as is.  Returns None for the first scan, then a dict with
        ``translation`` [3], ``quaternion`` (x,y,z,w), ``transformation`` [4,4] (previous -> current, metres) and, when
        integrating, ``global_translation`` / ``global_quaternion`` / ``T_0_t``."""
        if not torch.is_tensor(scan):
            scan = torch.from_numpy(filter_scans(np.asarray(scan, dtype=np.float32)))
        current = scan[:, :3].to(self.device).float().clone()
        result = None
        if self.point_cloud_t_1 is not None:
            previous = self.point_cloud_t_1
            if self.config["normalization_scaling"]:
                self.scaling_factor = self.normalize_input(input_1=current, input_2=previous)
            images = self.project_pair(previous, current, self.sensor)
            stacked = torch.cat((images[0:1], images[1:2]), dim=1)                # image_1 = previous, image_2 = current (:143)
            translation_1, rot_repr_1 = self.model(stacked)
            T = self.geometry_handler.get_transformation_matrix_quaternion(
                translation=translation_1, quaternion=rot_repr_1, device=self.device)
            quaternion = quaternion_from_matrix(T[0].double().cpu().numpy())
            translation = translation_1[0].double().cpu().numpy() * self.scaling_factor
            transformation = T[0].double().cpu().numpy().copy()
            transformation[:3, 3] = translation
            result = {"translation": translation, "quaternion": quaternion, "transformation": transformation}
            if self.integrator is not None:
                gt, gq = self.integrator.update_transformation(quaternion=quaternion, translation=translation)
                result.update(global_translation=gt, global_quaternion=gq, T_0_t=self.integrator.T_0_t[0].copy())
        self.point_cloud_t_1 = current * self.scaling_factor                    # undo the in-place scaling (:172)
        return result
