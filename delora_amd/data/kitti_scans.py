"""Raw KITTI odometry scans: ``<data_path>/<seq:02d>/velodyne/<idx:06d>.bin``, float32 ``[N,4]`` (x,y,z,intensity).
The reference goes through pykitti (src/data/kitti_scans.py:25-50); the file format needs nothing more than numpy."""
import glob
import os

import numpy as np
import torch


class KITTIDatasetPreprocessor:
    def __init__(self, config, dataset_name, preprocessing_fct):
        self.config, self.identifier, self.preprocessing_fct = config, dataset_name, preprocessing_fct

    def scan_files(self):
        seq = format(self.config[self.identifier]["data_identifier"], "02d")
        return sorted(glob.glob(os.path.join(self.config[self.identifier]["data_path"], seq, "velodyne", "*.bin")))

    def preprocess(self):
        device = self.config["device"]
        for index, path in enumerate(self.scan_files()):
            raw = np.fromfile(path, dtype=np.float32).reshape(-1, 4)
            scan = torch.from_numpy(np.ascontiguousarray(raw.T)).unsqueeze(0).to(device)     # [1,4,N] as kitti_scans.py:46-50
            self.preprocessing_fct(scan=scan, index=index)
