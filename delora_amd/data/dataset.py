"""Datasets of consecutive scan pairs.

``PreprocessedPointCloudDataset`` reads the reference's on-disk format
(``<preprocessed_path>/<seq:02d>/{scans,normals}/<idx:06d>.npy``, ``[M,3]`` fp32, written by
src/preprocessing/preprocesser.py:64-68) and yields the same sample dict as src/data/dataset.py:124-154.
``SyntheticPairDataset`` produces pairs from delora_amd.data.synthetic when no dataset is on disk
(build container, bench); its samples carry raw scans and, unless asked otherwise, no normal lists --
the step then estimates normals online from the projected images.
"""
import glob
import os

import numpy as np
import torch

from . import synthetic


def _as_list_tensor(array):
    """``[M,3]`` array -> ``[1,3,M]`` tensor (the reference's view of a stored list, dataset.py:94-102)."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(array, dtype=np.float32).T)).unsqueeze(0)


class PreprocessedPointCloudDataset(torch.utils.data.Dataset):
    """The reference's dataset of consecutive scan pairs over its on-disk format (src/data/dataset.py:19-157).  Additions for feeding a
    GPU at rate: a sequence directory WITHOUT ``normals/`` (or config ``load_normal_lists: false``) is an xyz-only tree -- the samples
    carry ``normal_list_* = None`` and the step estimates the normals online (half the bytes per pair); the scan shared by two
    consecutive samples (k+1 of one pair = k of the next) is decoded once when samples are visited in order (a two-entry cache per
    worker process)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.store_dataset_in_RAM = config["store_dataset_in_RAM"]
        self.load_normals = bool(config.get("load_normal_lists", True))
        self.scans_files_in_datasets, self.normals_files_in_datasets = [], []
        self._recent = {}                            # (dataset, sequence, scan) -> (normals, scan): the last two decoded scans
        table = []                                   # rows: (dataset index, sequence index, scan index)
        for i_ds, name in enumerate(config["datasets"]):
            scans_seq, normals_seq = [], []
            for i_seq, ident in enumerate(config[name]["data_identifiers"]):
                root = os.path.join(config[name]["preprocessed_path"], format(ident, "02d") + "/")
                if not os.path.exists(root):
                    raise Exception("The specified path and dataset " + root + "does not exist.")
                scans = sorted(glob.glob(os.path.join(root, "scans/", "*.npy")))
                normals = sorted(glob.glob(os.path.join(root, "normals/", "*.npy"))) if self.load_normals else []
                if self.load_normals and not normals and not os.path.isdir(os.path.join(root, "normals")):
                    # an xyz-only tree.  The mode holds for the WHOLE dataset and is decided by the first sequence: a later sequence
                    # without normals/ next to earlier ones that have them would silently drop stored normals (advisor, round 4)
                    if table:
                        raise Exception("The sequence " + root + " has no normals/ directory but earlier sequences do: mixed trees are "
                                        "not supported (set load_normal_lists: false to train every sequence with online normals).")
                    self.load_normals = False
                    print("Dataset: " + root + " holds no stored normal lists -- xyz-only mode, normals are estimated online for every sequence.")
                if not self.load_normals and bool(config.get("load_normal_lists", True)) and os.path.isdir(os.path.join(root, "normals")) \
                        and glob.glob(os.path.join(root, "normals/", "*.npy")):
                    raise Exception("The sequence " + root + " holds stored normal lists but an earlier sequence does not: mixed trees "
                                    "are not supported (set load_normal_lists: false to ignore them).")
                if self.load_normals and len(normals) != len(scans):
                    raise Exception("The sequence " + root + " holds " + str(len(scans)) + " scans but " + str(len(normals)) + " normal lists.")
                scans_seq.append(scans)
                normals_seq.append(normals)
                # a sample is the pair (t, t+1): the last scan of a sequence only ever appears as "t+1"
                table += [(i_ds, i_seq, k) for k in range(len(scans) - 1)]
            self.scans_files_in_datasets.append(scans_seq)
            self.normals_files_in_datasets.append(normals_seq)
        table = np.asarray(table, dtype=int).reshape(-1, 3)
        self.indices_dataset, self.indices_sequence, self.indices_scan = table[:, 0], table[:, 1], table[:, 2]
        self.num_scans_overall = len(table)
        self._ram = {}
        if self.store_dataset_in_RAM:
            print("Loading all scans and normals into the RAM / SWAP... Disable this if you do not have enough RAM.")
            for i_ds, seqs in enumerate(self.scans_files_in_datasets):
                for i_seq, files in enumerate(seqs):
                    for k in range(len(files)):
                        self._ram[(i_ds, i_seq, k)] = self.load_files_from_disk(i_ds, i_seq, k)
            print("Loaded " + str(len(self._ram)) + " scans to RAM/swap.")
        else:
            print("Dataset will be kept on disk. For higher performance enable RAM loading.")

    def load_files_from_disk(self, index_dataset, index_sequence, index_scan):
        normals = None
        if self.load_normals:
            normals = _as_list_tensor(np.load(self.normals_files_in_datasets[index_dataset][index_sequence][index_scan]))
        scan = _as_list_tensor(np.load(self.scans_files_in_datasets[index_dataset][index_sequence][index_scan]))
        return normals, scan

    def _get(self, i_ds, i_seq, k):
        if self.store_dataset_in_RAM:
            return self._ram[(i_ds, i_seq, k)]
        key = (i_ds, i_seq, k)
        hit = self._recent.get(key)
        if hit is None:
            hit = self.load_files_from_disk(i_ds, i_seq, k)
            if len(self._recent) >= 2:
                self._recent.pop(next(iter(self._recent)))
            self._recent[key] = hit
        return hit

    # -- raw access for data.feed.PackedFeed (worker processes write the files straight into the step's planar batch buffer)
    def _read_npy_into(self, path, pool):
        """The ``[M,3]`` fp32 array of an .npy file, read with ``readinto`` into one of this process's persistent scratch buffers: no
        allocation per file.  (np.load allocates -- and the kernel zero-fills and later unmaps -- 1.7 MB per scan: at 3000 scans/s the
        page faults of six worker processes slowed every process of the container down, the training thread included.)"""
        import numpy.lib.format as fmt
        with open(path, "rb") as f:
            major, _ = fmt.read_magic(f)
            shape, fortran, dtype = fmt.read_array_header_1_0(f) if major == 1 else fmt.read_array_header_2_0(f)
            if fortran or dtype != np.float32 or len(shape) != 2:
                f.seek(0)
                return np.ascontiguousarray(np.load(f), dtype=np.float32)
            count = int(shape[0]) * int(shape[1])
            slot = pool["next"] = (pool.get("next", -1) + 1) % 4            # four buffers: two cached scans + the two being read
            buf = pool.get(slot)
            if buf is None or buf.size < count:
                buf = pool[slot] = np.empty((max(count, 3 * self.max_points_per_scan()),), dtype=np.float32)
            view = buf[:count]
            got = f.readinto(memoryview(view).cast("B"))
            if got != count * 4:
                raise IOError(f"{path}: short read ({got} of {count * 4} bytes)")
            return view.reshape(shape)

    def _arrays(self, i_ds, i_seq, k):
        key = ("raw", i_ds, i_seq, k)
        hit = self._recent.get(key)
        if hit is None:
            pools = self.__dict__.setdefault("_scratch", ({}, {}))
            xyz = self._read_npy_into(self.scans_files_in_datasets[i_ds][i_seq][k], pools[0])
            nrm = self._read_npy_into(self.normals_files_in_datasets[i_ds][i_seq][k], pools[1]) if self.load_normals else None
            hit = (xyz, nrm)
            if len(self._recent) >= 2:
                self._recent.pop(next(iter(self._recent)))
            self._recent[key] = hit
        return hit

    def load_pair_arrays(self, index):
        """[(xyz, normals | None) of scan k, the same of scan k+1] without tensors: ``[M,3]`` arrays as stored on disk, or -- with
        ``store_dataset_in_RAM`` -- the ``[3,M]`` planar views of the tensors held in RAM (``PackedFeed`` tells the two layouts apart by
        their shape and copies either into the planar batch slot)."""
        i_ds, i_seq, k = int(self.indices_dataset[index]), int(self.indices_sequence[index]), int(self.indices_scan[index])
        if self.store_dataset_in_RAM:
            out = []
            for kk in (k, k + 1):
                normals, scan = self._ram[(i_ds, i_seq, kk)]
                out.append((scan[0].numpy(), normals[0].numpy() if normals is not None else None))
            return out
        return [self._arrays(i_ds, i_seq, k), self._arrays(i_ds, i_seq, k + 1)]

    def max_points_per_scan(self):
        """Upper bound of a stored list's length: the preprocessing image of its dataset block (vertical_cells x
        horizontal_cells_preprocessing, src/preprocessing/preprocesser.py:50-61), or config ``feed_points_per_scan``."""
        if self.config.get("feed_points_per_scan"):
            return int(self.config["feed_points_per_scan"])
        best = 0
        for name in self.config["datasets"]:
            block = self.config[name]
            best = max(best, int(block["vertical_cells"]) * int(block.get("horizontal_cells_preprocessing", block["horizontal_cells"])))
        return best

    def __getitem__(self, index):
        i_ds, i_seq, k = int(self.indices_dataset[index]), int(self.indices_sequence[index]), int(self.indices_scan[index])
        normal_list_1, scan_1 = self._get(i_ds, i_seq, k)
        normal_list_2, scan_2 = self._get(i_ds, i_seq, k + 1)
        return {"index": index, "index_dataset": i_ds, "index_sequence": i_seq, "index_scan": k,
                "dataset": self.config["datasets"][i_ds],
                "normal_list_1": normal_list_1, "normal_list_2": normal_list_2, "scan_1": scan_1, "scan_2": scan_2}

    def __len__(self):
        return self.num_scans_overall


class SyntheticPairDataset(torch.utils.data.Dataset):
    """``length`` deterministic synthetic pairs for dataset block ``dataset`` (seed = base_seed + index)."""

    def __init__(self, config, dataset, length, base_seed=1000, rings=None, azimuth_steps=None, points=None):
        self.config, self.dataset, self.length, self.base_seed = config, dataset, int(length), int(base_seed)
        block = config[dataset]
        self.rings = rings or block["vertical_cells"]
        self.azimuth_steps = azimuth_steps or block.get("horizontal_cells_preprocessing", block["horizontal_cells"])
        vf = block["vertical_field_of_view"]
        self.vfov_deg = (float(np.rad2deg(vf[0])), float(np.rad2deg(vf[1])))
        self.points = points

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        s1, s2, T = synthetic.make_pair(self.base_seed + index, rings=self.rings, azimuth_steps=self.azimuth_steps,
                                        vfov_deg=self.vfov_deg)
        if self.points:
            s1, s2 = synthetic.pad_or_trim(s1, self.points), synthetic.pad_or_trim(s2, self.points)
        return {"index": index, "index_dataset": 0, "index_sequence": 0, "index_scan": index, "dataset": self.dataset,
                "scan_1": torch.from_numpy(s1).unsqueeze(0), "scan_2": torch.from_numpy(s2).unsqueeze(0),
                "normal_list_1": None, "normal_list_2": None, "T_true": torch.from_numpy(T)}


class ListDataset(torch.utils.data.Dataset):
    """A dataset over an in-memory list of sample dicts (bench, smoke test, unit tests)."""

    def __init__(self, samples):
        self.samples = list(samples)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        return dict(self.samples[index])
