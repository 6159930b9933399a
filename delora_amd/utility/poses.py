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


def relative_pose_errors(poses_estimated, poses_ground_truth, lengths_m=(100.0, 200.0, 300.0, 400.0, 500.0, 600.0, 700.0, 800.0), step=10):
    """Relative pose error of an integrated trajectory in the style of the KITTI odometry benchmark, the protocol the reference's
    results are quoted in (the reference itself only writes the KITTI pose file, src/utility/poses.py:67-74, and leaves the scoring
    to the benchmark's development kit, which is third party and absent here; restated from its published definition): for every
    ``step``-th start frame i and every segment length L, j is the first frame whose travelled ground-truth distance from i reaches
    L; the error transform is ``inv(inv(E_i) E_j) @ (inv(G_i) G_j)``; its translation norm / L is the translation error (a
    fraction: x100 = percent) and its rotation angle / L the rotation error (rad/m).  Returns ``{"translation": mean fraction,
    "rotation_rad_per_m": mean, "rotation_deg_per_100m": ..., "segments": count, "per_length": {L: (t, r, n)}}``; the means run over
    all segments, as the development kit's do.  ``lengths_m`` is free because synthetic sequences are tens of metres long."""
    E = np.asarray(poses_estimated, dtype=np.float64).reshape(-1, 4, 4)
    G = np.asarray(poses_ground_truth, dtype=np.float64).reshape(-1, 4, 4)
    if E.shape != G.shape:
        raise ValueError(f"trajectories differ in length: {E.shape[0]} vs {G.shape[0]} poses")
    dist = np.concatenate(([0.0], np.cumsum(np.linalg.norm(np.diff(G[:, :3, 3], axis=0), axis=1))))
    per_length, t_all, r_all = {}, [], []
    for L in lengths_m:
        t_err, r_err = [], []
        for i in range(0, len(G), max(1, int(step))):
            j = int(np.searchsorted(dist, dist[i] + L, side="left"))
            if j >= len(G):
                break
            err = np.linalg.inv(np.linalg.inv(E[i]) @ E[j]) @ (np.linalg.inv(G[i]) @ G[j])
            cos = max(-1.0, min(1.0, 0.5 * (np.trace(err[:3, :3]) - 1.0)))
            t_err.append(np.linalg.norm(err[:3, 3]) / L)
            r_err.append(np.arccos(cos) / L)
        if t_err:
            per_length[float(L)] = (float(np.mean(t_err)), float(np.mean(r_err)), len(t_err))
            t_all += t_err
            r_all += r_err
    if not t_all:
        return {"translation": float("nan"), "rotation_rad_per_m": float("nan"), "rotation_deg_per_100m": float("nan"), "segments": 0,
                "per_length": {}}
    return {"translation": float(np.mean(t_all)), "rotation_rad_per_m": float(np.mean(r_all)),
            "rotation_deg_per_100m": float(np.degrees(np.mean(r_all)) * 100.0), "segments": len(t_all), "per_length": per_length}
