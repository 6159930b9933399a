"""``NormalsComputer``: per-dataset normal estimation with the reference's interface
(src/preprocessing/normal_computation.py:13-122) on top of the HIP stencil kernel."""
import torch

from .. import geometry


class NormalsComputer:
    def __init__(self, config, dataset_name):
        self.config = config
        self.dataset_name = dataset_name

    def _params(self):
        side = self.config[self.dataset_name]["neighborhood_side_length"]
        return (int(side[0] / 2), int(side[1] / 2), float(self.config["epsilon_range"]),
                int(self.config["min_num_points_in_neighborhood_to_determine_point_class"]))

    def compute_normal_image(self, image):
        """Normals as an image ``[S,3,H,W]`` (zero vector = none) for ``image[S,>=3,H,W]``; what the training step uses."""
        a, b, eps, min_n = self._params()
        return geometry.normals(image, a, b, eps, min_n)

    def compute_normal_vectors(self, image):
        """Reference return value (normal_computation.py:87): (normals ``[M,3]`` with zeros where none,
        has_normal ``[M]`` bool, points ``[M,3]``) over the valid pixels (x,y,z all non-zero) in raster order."""
        img = image[:1, :3].contiguous().float()
        nrm = self.compute_normal_image(img)[0]
        flat = img[0].reshape(3, -1)
        valid = (flat[0] != 0) & (flat[1] != 0) & (flat[2] != 0)
        normals = nrm.reshape(3, -1)[:, valid].transpose(0, 1).contiguous()
        points = flat[:, valid].transpose(0, 1).contiguous()
        has = (normals != 0).any(dim=1)
        return normals, has, points
