"""Geometry of one training step, batched: every sample of the batch goes through ONE launch per kernel.

The reference loops over the samples in Python and crosses to the host several times per sample
(src/deploy/deployer.py:245-268, :290-327).  Here the 2B scans of a batch are concatenated (CSR offsets),
projected together into a ``[2B,4,H,W]`` buffer laid out as [b0.scan_1, b0.scan_2, b1.scan_1, ...] -- which
*is* the network input ``[B,8,H,W]`` without a copy -- and normals, correspondences and losses are batched
the same way.  Nothing in here synchronises with the host.
"""
import torch

from .. import geometry


class PackedBatch:
    """A batch whose 2B scans already lie concatenated in ONE static device buffer (deploy/graph_step.py): ``pts [C, capacity]``
    fp32 (C = 3, or 6 with the stored normals as extra rows), CSR ``offs [2B+1]`` int32 on the device in the order b0.scan_1,
    b0.scan_2, b1.scan_1 ..., ``max_points`` = the per-scan capacity the launch is sized for.  Shapes and addresses never change
    between steps, only the contents and the offsets -- which is what lets a captured HIP graph run ragged batches."""

    def __init__(self, pts, offs, max_points, batch_size, dataset, with_lists):
        self.pts, self.offs, self.max_points, self.B, self.dataset, self.with_lists = pts, offs, int(max_points), int(batch_size), dataset, bool(with_lists)

    def __len__(self):
        return self.B


class HipStepGeometry:
    """Default (and only product) backend; raises through delora_amd._lib when libdelora_hip.so is missing."""

    def __init__(self):
        self._offsets = {}

    def _offsets_for(self, lengths, device):
        key = (tuple(lengths), str(device))
        t = self._offsets.get(key)
        if t is None:
            offs = [0]
            for n in lengths:
                offs.append(offs[-1] + n)
            host = torch.tensor(offs, dtype=torch.int32)
            if getattr(device, "type", str(device)) == "cuda":
                # through page-locked memory, asynchronously: a copy from pageable memory makes the host wait for everything queued on
                # the stream -- with ragged batches (every step a new key) the host could never run ahead of the GPU
                host = host.pin_memory()
                t = (host.to(device, non_blocking=True), host)            # (the staging vector lives as long as the cache entry)
            else:
                t = (host.to(device), None)
            if len(self._offsets) > 64:
                # drop the OLDEST entry only: its copy was issued 64 steps ago; the staging vectors of the latest steps may still be read
                self._offsets.pop(next(iter(self._offsets)))
            self._offsets[key] = t
        return t[0]

    def prepare(self, samples, sensor, normal_params):
        """samples: list of dicts with ``scan_1``/``scan_2`` ``[1,3,N]`` on the GPU and optional ``normal_list_*``.
        Returns dict(stacked [B,8,H,W] network input, images [B,2,4,H,W] view, normals [B,2,3,H,W], their packed twins
        packed / normals_packed [B,2,H,W,4], pix2pt [B,2,H,W])."""
        B = len(samples)
        if isinstance(samples, PackedBatch):
            with_lists = samples.with_lists
            out = geometry.project(samples.pts, samples.offs, samples.max_points, sensor, want_kept=False)
        else:
            with_lists = samples[0].get("normal_list_1") is not None
            chunks, lengths = [], []
            for s in samples:
                for k in ("1", "2"):
                    scan = s["scan_" + k][0]
                    if with_lists:
                        scan = torch.cat((scan[:3], s["normal_list_" + k][0]), dim=0)
                    chunks.append(scan)
                    lengths.append(scan.shape[1])
            pts = torch.cat(chunks, dim=1).contiguous().float()
            offs = self._offsets_for(lengths, pts.device)
            out = geometry.project(pts, offs, max(lengths), sensor, want_kept=False)
        image4 = out["image4"]
        H, W = sensor.H, sensor.W
        if with_lists:
            normals = out["aux"][:, :3]                 # stored normals ride along as extra channels (deployer.py:258-261)
            if normals.shape[1] != 3 or not normals.is_contiguous():
                normals = normals.contiguous()
            normals_pk = out["packed_aux"]
        else:
            a, b, eps, min_n = normal_params
            normals, normals_pk = geometry.normals(image4, a, b, eps, min_n, want_packed=True)
        return {"stacked": image4.view(B, 8, H, W), "images": image4.view(B, 2, 4, H, W),
                "normals": normals.view(B, 2, 3, H, W), "packed": out["packed"].view(B, 2, H, W, 4),
                "normals_packed": normals_pk.view(B, 2, H, W, 4), "pix2pt": out["pix2pt"].view(B, 2, H, W),
                "sensor": sensor}

    def losses(self, T, prepared, flags, need_without_normals):
        """(loss_terms [B,3] differentiable w.r.t. T, pair_counts [B,2], visible [B]) for target = scan 1, source = scan 2."""
        sensor = prepared["sensor"]
        src, src_n = prepared["images"][:, 1], prepared["normals"][:, 1]            # streamed: planar
        tgt_pk, tgt_n_pk = prepared["packed"][:, 0], prepared["normals_packed"][:, 0]   # gathered (by the search): packed
        nn, visible, match = geometry.nn_correspond(src, src_n, tgt_pk, tgt_n_pk, T, sensor,
                                                    need_without_normals=need_without_normals)
        terms, counts = geometry.icp_loss(T, src, src_n, match, nn, flags)
        return terms, counts, visible
