"""``ICPLosses``: the reference's list-based loss module (src/losses/icp_losses.py:10-158).

The training step does not go through this class (it uses the fused image-based kernels of
delora_amd.geometry); it exists so that code written against the reference's list interface keeps working:
correspondences come from the exhaustive HIP nearest-neighbour kernel, the loss arithmetic on the gathered
pairs is a handful of torch ops so that gradients flow into the (already transformed) inputs as they do in
the reference."""
import torch

from .. import geometry


class ICPLosses(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if config["normal_loss"] not in ("squared", "linear"):
            raise Exception("The normal loss which is defined here is not admissible.")

    _warned_quadratic = False

    @staticmethod
    def find_target_correspondences(target_point_cloud, source_point_cloud):
        """Index of the exact nearest target point for every source point (both ``[1,3,M]``).  The list interface has no image lattice to
        search on, so this is the exhaustive O(M^2) kernel: 17 G distance evaluations per call for two 64x2048 scans.  Said once, loudly,
        above 32 k points -- the training step (``Deployer.step``) uses the lattice search of ``geometry.nn_correspond`` instead."""
        n_src, n_tgt = int(source_point_cloud.shape[-1]), int(target_point_cloud.shape[-1])
        if max(n_src, n_tgt) > 32768 and not ICPLosses._warned_quadratic:
            ICPLosses._warned_quadratic = True
            import warnings
            warnings.warn(f"ICPLosses (list interface): exhaustive nearest-neighbour search over {n_src} x {n_tgt} points "
                          f"({n_src * n_tgt / 1e9:.1f} G distance evaluations per call); the image-based step path "
                          "(Deployer.step / geometry.nn_correspond) is exact as well and ~100x faster", RuntimeWarning, stacklevel=3)
        return geometry.nn_bruteforce(source_point_cloud[0], target_point_cloud[0]).long()

    def forward(self, source_point_cloud_transformed, source_normal_list_transformed, target_point_cloud,
                target_normal_list, compute_pointwise_loss_bool):
        cfg = self.config
        dev = target_point_cloud.device
        zero = torch.zeros(1, device=dev)
        losses = {"loss_po2po": zero, "loss_po2pl": zero, "loss_po2pl_pointwise": zero, "loss_pl2pl": zero}
        if cfg["po2po_alone"]:                                             # icp_losses.py:36-45
            if cfg["point_to_plane_loss"] or cfg["plane_to_plane_loss"]:       # the reference dies here (:135-146)
                raise Exception("po2po_alone needs point_to_plane_loss and plane_to_plane_loss switched off.")
            nn = self.find_target_correspondences(target_point_cloud, source_point_cloud_transformed)
            if cfg["point_to_point_loss"]:
                d = source_point_cloud_transformed - target_point_cloud[:, :, nn]
                losses["loss_po2po"] = (d * d).mean()
            return losses, None
        s_has = (source_normal_list_transformed[0] != 0).any(dim=0)          # :48-50
        t_has = (target_normal_list[0] != 0).any(dim=0)                      # :51-52
        src_w = source_point_cloud_transformed[:, :, s_has]
        srcn_w = source_normal_list_transformed[:, :, s_has]
        nn_w = self.find_target_correspondences(target_point_cloud, src_w)
        if cfg["point_to_point_loss"]:                                     # :85-100: neither side has a normal
            src_wo = source_point_cloud_transformed[:, :, ~s_has]
            nn_wo = self.find_target_correspondences(target_point_cloud, src_wo)
            keep = ~t_has[nn_wo]
            d = src_wo[:, :, keep] - target_point_cloud[:, :, nn_wo[keep]]
            losses["loss_po2po"] = (d * d).mean()
        keep = t_has[nn_w]                                                 # :110-121: target must have a normal too
        s_k, sn_k = src_w[:, :, keep], srcn_w[:, :, keep]
        t_k, tn_k = target_point_cloud[:, :, nn_w[keep]], target_normal_list[:, :, nn_w[keep]]
        if cfg["point_to_plane_loss"]:                                     # :196-203
            dvec = s_k - t_k
            r = (dvec * tn_k).sum(dim=1)
            losses["loss_po2pl"] = (r * r).mean()
            if compute_pointwise_loss_bool:
                losses["loss_po2pl_pointwise"] = dvec
        if cfg["plane_to_plane_loss"]:                                     # :224-238
            if cfg["normal_loss"] == "linear":
                c = (sn_k * tn_k).sum(dim=1)
                losses["loss_pl2pl"] = ((1 - c) ** 2).mean()
            else:
                e = sn_k - tn_k
                losses["loss_pl2pl"] = (e * e).sum(dim=1).mean()
        plotting = {"scan_2_transformed": s_k, "normals_2_transformed": sn_k}
        return losses, plotting
