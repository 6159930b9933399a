"""``Deployer``: model + geometry + one optimisation step over a batch of scan pairs.

Mirror of the reference's Deployer (src/deploy/deployer.py:21-375) for the training/inference step:
same constructor argument, same ``step(preprocessed_dicts, epoch_losses, log_images_bool)`` contract and
the same loss bookkeeping -- including its accumulation order, which gives sample j of a batch the weight
(B-j)/B in the back-propagated ``loss_pc`` (deployer.py:309-312,329) -- but the per-sample Python loops are
replaced by batched HIP launches (deploy/step_geometry.py) and nothing synchronises with the host inside a step.
Plotting / MLflow image logging (deployer.py:73-162) is outside the hot path and not provided.
"""
import torch

from .. import geometry
from ..data import dataset as dataset_module
from ..models import model as model_module
from ..models import model_parts
from ..utility import projection
from . import step_geometry


class Deployer(object):

    def __init__(self, config, dataset=None, geometry_backend=None):
        self.config = config
        self.device = config["device"]
        self.batch_size = config["batch_size"]
        self.dataset = dataset if dataset is not None else dataset_module.PreprocessedPointCloudDataset(config=config)
        self.steps_per_epoch = int(len(self.dataset) / self.batch_size)
        self.img_projection = projection.ImageProjectionLayer(config=config)
        self.model = model_module.OdometryModel(config=config).to(self.device)
        if config.get("channels_last", False):
            self.model = self.model.to(memory_format=torch.channels_last)
        elif getattr(self.device, "type", "cpu") == "cuda" and config.get("cnn_impl", "auto") != "modules":
            # the HIP trunk reads the parameters in channels_last storage in place.  Only worth it -- and only harmless --
            # when every configured image size actually takes that path: the module path (library convolutions on NCHW
            # activations) would re-lay-out channels_last weights or activations on every call.
            sizes = [(config[d]["vertical_cells"], config[d]["horizontal_cells"]) for d in config["datasets"]]
            if not config.get("pre_feature_extraction", False) and all(
                    self.model.resnet.hip_path_takes(H, W) for (H, W) in sizes):
                self.model.resnet.trunk_weights_channels_last()
        if config["use_jit"]:
            first = config["datasets"][0]
            example = torch.zeros((1, 4, config[first]["vertical_cells"], config[first]["horizontal_cells"]), device=self.device)
            self.model = torch.jit.trace(self.model, example_inputs=(example, example))
        self.geometry_handler = model_parts.GeometryHandler(config=config)
        self.geo = geometry_backend if geometry_backend is not None else step_geometry.HipStepGeometry()
        self.lossTransformation = torch.nn.MSELoss()
        self.training_bool = False
        # float16 autocast: the trunk's inter-layer gradients are STORED in fp16 and start at (head gradient) / (H*W/64) -- below
        # fp16's normal range (6e-5) for O(1e-2) loss gradients -- so the backward runs on a scaled loss (dynamic scale, skipped steps
        # on overflow: torch's GradScaler, which hands scale and overflow flag to the fused Adam kernel without a host sync).
        # bfloat16 has fp32's exponent range and needs none.
        self.grad_scaler = None
        if config.get("amp_dtype") == "float16" and getattr(self.device, "type", "cpu") == "cuda" and config.get("amp_loss_scaling", True):
            self.grad_scaler = torch.amp.GradScaler("cuda", init_scale=float(config.get("amp_init_scale", 1024.0)))
        # data-parallel placement of this process' slice inside the global batch (single process: 0 of 1)
        self.rank, self.world_size = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world_size = torch.distributed.get_rank(), torch.distributed.get_world_size()

    @staticmethod
    def list_collate(batch_dicts):
        return list(batch_dicts)

    # ------------------------------------------------------------------ point cloud helpers (deployer.py:181-189)
    def rotate_point_cloud_transformation_matrix(self, transformation_matrix, point_cloud):
        return transformation_matrix[:, :3, :3].matmul(point_cloud[:, :3, :])

    def transform_point_cloud_transformation_matrix(self, transformation_matrix, point_cloud):
        return (self.rotate_point_cloud_transformation_matrix(transformation_matrix, point_cloud)
                + transformation_matrix[:, :3, 3].view(-1, 3, 1))

    def augment_input(self, preprocessed_data):
        if self.config["random_point_cloud_rotations"]:
            raise Exception("Needs to be verified for larger batches")   # the reference refuses too (deployer.py:203-204)
        return preprocessed_data

    def normalize_input(self, preprocessed_data):
        """Optional range normalisation (deployer.py:222-235): both scans divided by the mean of their mean ranges."""
        m1 = torch.norm(preprocessed_data["scan_1"], dim=1).mean(dim=1, keepdim=True)
        m2 = torch.norm(preprocessed_data["scan_2"], dim=1).mean(dim=1, keepdim=True)
        mean = torch.cat((m1, m2), dim=1).mean(dim=1)
        preprocessed_data["scan_1"] = preprocessed_data["scan_1"] / mean
        preprocessed_data["scan_2"] = preprocessed_data["scan_2"] / mean
        preprocessed_data["scaling_factor"] = mean
        return preprocessed_data, mean

    def _normal_params(self, dataset):
        side = self.config[dataset]["neighborhood_side_length"]
        return (int(side[0] / 2), int(side[1] / 2), float(self.config["epsilon_range"]),
                int(self.config["min_num_points_in_neighborhood_to_determine_point_class"]))

    def _run_model(self, stacked):
        amp = self.config.get("amp_dtype")
        if self.config.get("channels_last", False):
            stacked = stacked.contiguous(memory_format=torch.channels_last)
        args = (stacked[:, :4], stacked[:, 4:]) if self.config["use_jit"] else (stacked,)
        if amp:
            with torch.autocast("cuda", dtype=getattr(torch, amp)):
                t, q = self.model(*args)
            return t.float(), q.float()
        return self.model(*args)

    # ------------------------------------------------------------------------------------------------ the step
    def _loss_weights(self, B, Bg, lam, device, dtype):
        """Cached constants of the loss weighting: W_pc[j,k] = (Bg - j_global) / Bg * (1, lambda, 1)[k] and the column
        weights (1, lambda, 1) / Bg of the logged per-term sums."""
        key = (B, Bg, self.rank, lam, str(device), dtype)
        cache = self.__dict__.setdefault("_loss_weight_cache", {})
        if key not in cache:
            j = torch.arange(B, dtype=torch.float64) + float(self.rank * B)
            col = torch.tensor([1.0, lam, 1.0], dtype=torch.float64)
            cache[key] = ((((Bg - j) / Bg)[:, None] * col[None, :]).to(dtype).to(device), (col / Bg).to(dtype).to(device))
        return cache[key]

    @staticmethod
    def _accumulate(epoch_losses, keys, vals):
        """epoch_losses[k] += v for 0-d device tensors.  While the accumulators are still the initial python zeros the values
        are stacked into ONE fresh vector whose elements become the accumulators (no aliasing of the step's own tensors);
        afterwards one multi-tensor add -- instead of one kernel per key and step."""
        if all(not torch.is_tensor(epoch_losses[k]) and epoch_losses[k] == 0.0 for k in keys):
            stacked = torch.stack([v.to(torch.float32) for v in vals])
            for i, k in enumerate(keys):
                epoch_losses[k] = stacked[i]
            return
        acc, add = [], []
        for k, v in zip(keys, vals):
            if torch.is_tensor(epoch_losses[k]):
                acc.append(epoch_losses[k])
                add.append(v.to(epoch_losses[k].dtype))
            else:
                epoch_losses[k] = epoch_losses[k] + v
        if acc and all(a.is_cuda for a in acc):
            torch._foreach_add_(acc, add)
        else:
            for a, v in zip(acc, add):
                a += v

    def step(self, preprocessed_dicts, epoch_losses=None, log_images_bool=False):
        cfg = self.config
        packed = preprocessed_dicts if isinstance(preprocessed_dicts, step_geometry.PackedBatch) else None
        B = len(preprocessed_dicts)
        if packed is not None:
            # a batch that already lies concatenated in static device buffers (deploy/graph_step.py): one sensor, no per-sample
            # host-side preprocessing
            if (self.training_bool and cfg["random_point_cloud_rotations"]) or cfg["normalization_scaling"]:
                raise ValueError("a PackedBatch cannot be augmented or range-normalised (config: random_point_cloud_rotations / normalization_scaling)")
            preprocessed_dicts = [{"dataset": packed.dataset}] * B
        if B != self.batch_size:
            # the reference indexes batch_size transforms against the list and fails on a short last batch
            # (deployer.py:240,290-292); make that explicit
            raise ValueError(f"step() needs exactly batch_size={self.batch_size} samples, got {B} (use drop_last)")
        for i, d in enumerate(preprocessed_dicts if packed is None else []):
            if self.training_bool:
                d = self.augment_input(preprocessed_data=d)
            if cfg["normalization_scaling"]:
                d, _ = self.normalize_input(preprocessed_data=d)
            preprocessed_dicts[i] = d
        # Samples are grouped by image geometry (sensor rings / columns / fields of view).  The reference requires ONE
        # geometry per batch (hyperparameters.yaml:3, deployer.py:240-243) -- then there is exactly one group and one launch
        # per kernel.  A mixed-sensor batch (BASELINE config 5; no reference semantics) runs one sub-batch per geometry
        # through projection / CNN / correspondences / loss and is re-assembled in sample order, so that every sample sees
        # exactly what it would see in a batch of its own kind and the loss weights still follow the sample index.
        groups = {}
        for i, d in enumerate(preprocessed_dicts):
            groups.setdefault(self.img_projection.sensor(d["dataset"]).key(), []).append(i)
        flags = geometry.loss_flags(cfg)
        T_rows, term_rows, count_rows, vis_rows = [None] * B, [None] * B, [None] * B, [None] * B
        for idx in groups.values():
            dataset = preprocessed_dicts[idx[0]]["dataset"]
            sensor = self.img_projection.sensor(dataset)
            prepared = self.geo.prepare(packed if packed is not None else [preprocessed_dicts[i] for i in idx], sensor,
                                        self._normal_params(dataset))
            translations, rotation_representation = self._run_model(prepared["stacked"])
            T_g = self.geometry_handler.get_transformation_matrix_quaternion(
                translation=translations, quaternion=rotation_representation, device=self.device)
            if not cfg["inference_only"]:
                terms_g, counts_g, vis_g = self.geo.losses(T_g, prepared, flags,
                                                           need_without_normals=geometry.need_without_normals(cfg))
            for k, i in enumerate(idx):
                T_rows[i] = T_g[k]
                if not cfg["inference_only"]:
                    term_rows[i] = terms_g[k]
                    count_rows[i] = counts_g[k] if counts_g is not None else None
                    vis_rows[i] = vis_g[k] if vis_g is not None else None
        single = len(groups) == 1
        computed_transformations = T_g if single else torch.stack(T_rows)

        def rescale(T):
            if cfg["normalization_scaling"]:
                T = T.clone()
                for i, d in enumerate(preprocessed_dicts):
                    T[i, :3, 3] = T[i, :3, 3] * d["scaling_factor"]
            return T

        if cfg["inference_only"]:
            return rescale(computed_transformations)

        terms = terms_g if single else torch.stack(term_rows)
        counts = counts_g if single else (torch.stack(count_rows) if count_rows[0] is not None else None)
        visible = vis_g if single else (torch.stack(vis_rows) if vis_rows[0] is not None else None)
        # global batch bookkeeping: this rank holds samples [rank*B, (rank+1)*B) of a batch of world_size*B.
        # lambda_po2pl scales the point-to-plane column only (deployer.py:309-311); the running sums added inside the sample
        # loop (:312) put the weight (Bg - j) on sample j, then everything is divided by Bg (:329).  Both are constants of the
        # run: one cached [B,3] weight matrix, so the weighted loss is a multiply and a sum (and two kernels in the backward)
        Bg = B * self.world_size
        W_pc, w_col = self._loss_weights(B, Bg, float(cfg["lambda_po2pl"]), terms.device, terms.dtype)
        with torch.no_grad():
            sums = (terms.detach() * w_col).sum(dim=0)                    # [po2po, lambda * po2pl, pl2pl] / Bg
        losses = {"loss_po2po": sums[0], "loss_po2pl": sums[1], "loss_pl2pl": sums[2], "loss_pc": (terms * W_pc).sum()}
        if not cfg["unsupervised_at_start"]:
            # identity pre-training (:324-338): the reference overwrites loss_transformation in every loop pass, so
            # only the LAST sample of the batch is fitted to the identity
            eye = torch.eye(4, device=self.device).view(1, 4, 4)
            loss = self.lossTransformation(input=computed_transformations[B - 1:B], target=eye) / Bg
            if self.world_size > 1 and self.rank != self.world_size - 1:
                loss = loss * 0.0
        else:
            loss = losses["loss_pc"]
        if self.training_bool:
            # DDP averages gradients over ranks; the reference's loss is a SUM over the global batch
            if self.grad_scaler is not None:
                self.grad_scaler.scale(loss * float(self.world_size)).backward()
                self.grad_scaler.step(self.optimizer)
                self.grad_scaler.update()
            else:
                (loss * float(self.world_size)).backward()
                self.optimizer.step()
        computed_transformations = rescale(computed_transformations)
        if epoch_losses is not None:
            keys = ["loss_epoch", "loss_point_cloud_epoch", "loss_po2po_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch"]
            vals = [loss.detach(), losses["loss_pc"].detach(), losses["loss_po2po"], losses["loss_po2pl"], losses["loss_pl2pl"]]
            if visible is not None:
                keys.append("visible_pixels_epoch")
                vals.append(visible[B - 1].detach())                      # last sample only (:349-352)
            self._accumulate(epoch_losses, keys, vals)
        self.last_step = {"loss_terms": terms.detach(), "pair_counts": counts, "losses": {k: v.detach() for k, v in losses.items()}}
        return epoch_losses, computed_transformations
