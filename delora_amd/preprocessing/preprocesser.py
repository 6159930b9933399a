"""Offline preprocessing: raw scans -> projected point lists + normal lists on disk, the format the training dataset
reads (mirror of src/preprocessing/preprocesser.py:19-103; the cv2/matplotlib preview is not provided).

Each scan is projected at ``horizontal_cells_preprocessing`` (dl_project), normals are estimated on that image
(dl_normals) and ``<preprocessed_path>/<seq:02d>/{scans,normals}/<idx:06d>.npy`` receive the ``[M,3]`` lists of the
valid pixels in raster order -- exactly what ``NormalsComputer.compute_normal_vectors`` returns (preprocesser.py:60-68).
On MI355X this is a fraction of a millisecond per scan; the reference's CPU path takes ~3 s.
"""
import os

import numpy as np

from ..data import kitti_scans
from ..utility import projection
from . import normal_computation


class Preprocesser:
    def __init__(self, config):
        self.config = config
        self.img_projection = projection.ImageProjectionLayer(config=config)
        self.normals_computer = None
        self.scans_name = self.normals_name = None

    @staticmethod
    def ensure_dir(file_path):
        os.makedirs(os.path.dirname(file_path), exist_ok=True)

    def apply_preprocessing_step(self, scan, index):
        image, _, _, _, _ = self.img_projection(input=scan[:, :3].contiguous(), dataset=self.config["dataset"])
        normals, _, point_list = self.normals_computer.compute_normal_vectors(image=image)
        np.save(os.path.join(self.normals_name, format(int(index), "06d") + ".npy"), normals.cpu().numpy())
        np.save(os.path.join(self.scans_name, format(int(index), "06d") + ".npy"), point_list.cpu().numpy())

    def preprocess_scans(self, scans, dataset_name, data_identifier=0):
        """The preprocessing step over raw scans that are already in memory (an iterable of ``[3|4,N]`` arrays / tensors): writes
        ``<preprocessed_path>/<data_identifier:02d>/{scans,normals}/<idx:06d>.npy`` exactly as ``preprocess_data`` does for a KITTI
        directory (synthetic sequences of the bench, the tests and the convergence run go through the same code as real scans)."""
        import torch
        block = self.config[dataset_name]
        self.config["dataset"] = dataset_name
        block["horizontal_cells"] = block["horizontal_cells_preprocessing"]                # preprocesser.py:73-74
        self.normals_computer = normal_computation.NormalsComputer(config=self.config, dataset_name=dataset_name)
        name = os.path.join(block["preprocessed_path"], format(int(data_identifier), "02d") + "/")
        self.normals_name, self.scans_name = os.path.join(name, "normals/"), os.path.join(name, "scans/")
        self.ensure_dir(self.normals_name)
        self.ensure_dir(self.scans_name)
        for index, scan in enumerate(scans):
            scan = torch.as_tensor(np.ascontiguousarray(scan) if isinstance(scan, np.ndarray) else scan)
            self.apply_preprocessing_step(scan=scan.reshape(1, scan.shape[-2], scan.shape[-1]).to(self.config["device"]), index=index)
        return name

    def preprocess_data(self):
        for dataset_name in self.config["datasets"]:
            block = self.config[dataset_name]
            self.config["dataset"] = dataset_name
            block["horizontal_cells"] = block["horizontal_cells_preprocessing"]            # preprocesser.py:73-74
            self.normals_computer = normal_computation.NormalsComputer(config=self.config, dataset_name=dataset_name)
            for data_identifier in block["data_identifiers"]:
                block["data_identifier"] = data_identifier
                name = os.path.join(block["preprocessed_path"], format(data_identifier, "02d") + "/")
                self.normals_name, self.scans_name = os.path.join(name, "normals/"), os.path.join(name, "scans/")
                self.ensure_dir(self.normals_name)
                self.ensure_dir(self.scans_name)
                if block["dataset_type"] == "kitti":
                    kitti_scans.KITTIDatasetPreprocessor(config=self.config, dataset_name=dataset_name,
                                                         preprocessing_fct=self.apply_preprocessing_step).preprocess()
                else:
                    raise Exception('Dataset type not yet supported. Currently only "kitti" is available (rosbag input needs ROS).')
