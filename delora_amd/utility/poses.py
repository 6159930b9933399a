"""Pose integration for evaluation: chain the predicted scan-to-scan transforms into a trajectory in the KITTI
camera ("world") convention and write KITTI-format pose files (reference src/utility/poses.py:11-74)."""
import csv

import numpy as np
import scipy.spatial.transform

# LiDAR frame (x forward, y left, z up) -> KITTI camera frame (x right, y down, z forward), poses.py:20-27
LIDAR_TO_WORLD = np.array([[0.0, -1.0, 0.0, 0.0],
                           [0.0, 0.0, -1.0, 0.0],
                           [1.0, 0.0, 0.0, 0.0],
                           [0.0, 0.0, 0.0, 1.0]])


def check_validity_so3(r):
    """det = 1 and R^T R = I within 1e-6 (poses.py:59-64)."""
    return bool(np.isclose(np.linalg.det(r), 1.0, atol=1e-6)) and bool(np.allclose(r.T @ r, np.eye(3), atol=1e-6))


def compute_poses(computed_transformations):
    """``[K+1,4,4]`` world poses (first = identity) from K transforms, each ``[1,4,4]`` or ``[4,4]``: the accumulated
    LiDAR pose is right-multiplied by every step, its rotation is re-projected onto SO(3) through a normalised quaternion
    after every step, and the result is conjugated into the world frame (poses.py:30-55)."""
    to_lidar = LIDAR_TO_WORLD.T
    pose_lidar = np.eye(4)
    poses = [np.eye(4)]
    for step in computed_transformations:
        pose_lidar = pose_lidar @ np.asarray(step, dtype=np.float64).reshape(4, 4)
        quat = scipy.spatial.transform.Rotation.from_matrix(pose_lidar[:3, :3]).as_quat()
        quat = quat / np.linalg.norm(quat)
        pose_lidar[:3, :3] = scipy.spatial.transform.Rotation.from_quat(quat).as_matrix()
        pose_world = LIDAR_TO_WORLD @ pose_lidar @ to_lidar
        if not check_validity_so3(pose_world[:3, :3]):
            raise Exception("Pose is not valid!")
        poses.append(pose_world)
    return np.stack(poses, axis=0)


def write_poses_to_text_file(file_name, poses):
    """One line per pose: the first 12 entries of the row-major 4x4 matrix, space separated (poses.py:67-74)."""
    with open(file_name, "w", newline="") as f:
        writer = csv.writer(f, delimiter=" ")
        for pose in poses:
            writer.writerow(np.asarray(pose).reshape(16)[:12])
