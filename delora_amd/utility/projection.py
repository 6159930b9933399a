"""``ImageProjectionLayer``: the reference's single-scan projection module (src/utility/projection.py:11-109)
on top of the batched HIP projection.  Same constructor, same call signature, same 5-tuple."""
import torch

from .. import geometry


class ImageProjectionLayer(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.device = config["device"]
        self.config = config
        self.horizontal_field_of_view = config["horizontal_field_of_view"]
        self._sensors = {}

    def sensor(self, dataset):
        """Resolved (cached) sensor of a dataset block; re-resolved when the config entries change
        (the offline preprocessing rewrites ``horizontal_cells`` in place, src/preprocessing/preprocesser.py:73-74)."""
        s = geometry.Sensor.from_config(self.config, dataset)
        cached = self._sensors.get(dataset)
        if cached is None or cached.key() != s.key():
            self._sensors[dataset] = s
            return s
        return cached

    def project_to_img(self, point_cloud, dataset):
        """point_cloud ``[1,C,N]`` -> (image ``[1,C+1,H,W]`` with range as last channel, u ``[1,N]``, v ``[1,N]``,
        point_cloud_indices ``[M]`` int64 into the input in ascending range, image_to_pointcloud_indices ``[1,M,2]``
        int64 (v,u)), exactly the reference's return value (projection.py:105-106).  u and v are those of ALL points in
        range-sorted order.  The ordering costs two device sorts that the training step itself never needs
        (it calls geometry.project directly)."""
        sensor = self.sensor(dataset)
        B, C, N = point_cloud.shape
        if B != 1:
            raise ValueError("ImageProjectionLayer projects one scan per call (as the reference does)")
        pts = point_cloud[0].detach().contiguous().float()
        offs = torch.tensor([0, N], dtype=torch.int32, device=pts.device)
        out = geometry.project(pts, offs, N, sensor, want_uv=True)
        H, W = sensor.H, sensor.W
        img4 = out["image4"][0]
        parts = [img4[:3]] + ([out["aux"][0]] if C > 3 else []) + [img4[3:4]]
        image = torch.cat(parts, dim=0).unsqueeze(0)
        # all points in ascending range (ties: lower index first, as a stable sort gives)
        order = torch.sort(out["uv"][2], stable=True).indices
        u = out["uv"][0][order].unsqueeze(0)
        v = out["uv"][1][order].unsqueeze(0)
        # kept points in the same order: winners sorted by (range, index) = two stable sorts, minor key first
        pix2pt = out["pix2pt"][0].reshape(-1)
        occ = torch.nonzero(pix2pt >= 0).reshape(-1)
        by_index = torch.sort(pix2pt[occ].long(), stable=True)
        occ_i, idx_i = occ[by_index.indices], by_index.values
        by_range = torch.sort(img4[3].reshape(-1)[occ_i], stable=True).indices
        occ_s, idx_s = occ_i[by_range], idx_i[by_range]
        vu = torch.stack((torch.div(occ_s, W, rounding_mode="floor"), occ_s % W), dim=1).unsqueeze(0)
        return image, u, v, idx_s, vu

    def forward(self, input, dataset):
        return self.project_to_img(point_cloud=input, dataset=dataset)
