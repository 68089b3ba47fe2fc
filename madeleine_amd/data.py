"""Input side of the pretrain step: the dataset/collate contract of the reference
(madeleine/datasets/wsi_dataset.py:14-99) and a pinned, double-buffered host->device stager that overlaps the per-step
H2D copy of `feats [B,M,N,D]` (512 MiB at config 2, 8.5 ms on PCIe Gen5) with the previous step's compute on a side
HIP stream.  Host glue: no numerics here.
"""
import os
from typing import Callable, Iterable, Optional, Sequence

import torch
from torch.utils.data import Dataset


def load_features(h5_path):
    """wsi_dataset.py:14-19: the `features` dataset of an h5 file, squeezed, as a float tensor.  Uses h5py when it is
    installed, otherwise the ctypes binding of the system's libhdf5 (madeleine_amd/h5io.py) -- this image has no h5py."""
    try:
        import h5py
    except ImportError:
        from . import h5io
        return torch.from_numpy(h5io.read_dataset(h5_path, "features").squeeze())
    with h5py.File(h5_path, 'r') as f:
        feats = f['features'][:].squeeze()
    return torch.as_tensor(feats, dtype=torch.float32)


class SlideDataset(Dataset):
    """Mirror of wsi_dataset.py:21-84: one item = {'feats': list of M tensors [N,D], 'modality_labels', 'slide_id'}.
    Present stain -> features read by `feature_loader(path)`; absent stain -> a zero bag (`torch.zeros([2, D])`, :66);
    every bag is resampled to `sample` tokens (randperm without replacement, or randint with replacement when the bag
    is shorter, :42-50)."""

    def __init__(self, dataset_name, csv_path, features_path, modalities, embedding_size=None, sample=-1, train=True,
                 feature_loader: Optional[Callable] = None, dataframe=None):
        self.dataset_name = dataset_name
        if dataframe is None:
            import pandas as pd
            dataframe = pd.read_csv(csv_path)
        self.dataframe = dataframe
        self.features_path = features_path
        self.modalities = modalities
        self.sample = sample
        self.train = train
        self.embedding_size = embedding_size
        self.feature_loader = feature_loader or load_features

    def __len__(self):
        return len(self.dataframe)

    def sample_n(self, feats):
        if self.sample > -1:
            if feats.shape[0] < self.sample:
                idx = torch.randint(0, feats.shape[0], (self.sample,))
            else:
                idx = torch.randperm(feats.shape[0])[:self.sample]
            feats = feats[idx]
        return feats

    def __getitem__(self, index):
        row = self.dataframe.iloc[index]
        slide_id = row['slide_id']
        labels = [row[m] for m in self.modalities]
        if self.train:
            split = row['split']
            special = "" if split == "train" else f"_{split}"
            feats = []
            for m, lab in zip(self.modalities, labels):
                path = os.path.join(self.features_path, f"{slide_id}_{m}{special}.h5")
                cur = self.feature_loader(path) if lab == 1 else torch.zeros([2, self.embedding_size])
                feats.append(self.sample_n(cur))
        else:
            feats = [self.feature_loader(os.path.join(self.features_path, f"{slide_id}.h5"))]
            labels = [1]
        return {'feats': feats, 'modality_labels': labels, 'slide_id': slide_id}


def collate(batch):
    """wsi_dataset.py:86-99 -> {'feats': [B,M,N,D], 'modality_labels': [B,M], 'slide_ids': list}."""
    return {"feats": torch.stack([torch.stack(item['feats']) for item in batch]),
            "modality_labels": torch.stack([torch.Tensor(item['modality_labels']) for item in batch]),
            "slide_ids": [item['slide_id'] for item in batch]}


class SimpleDataset(Dataset):
    """Mirror of wsi_dataset.py:102-120 (the extraction-side dataset run_inference iterates): every `*.h5` file of a directory ->
    (features [N, D] float tensor, slide id = file name without the extension)."""

    def __init__(self, features_path, feature_loader: Optional[Callable] = None):
        self.features_path = features_path
        self.fnames = sorted(fn for fn in os.listdir(features_path) if fn.endswith('.h5'))
        self.feature_loader = feature_loader or load_features

    def __len__(self):
        return len(self.fnames)

    def __getitem__(self, index):
        feats = self.feature_loader(os.path.join(self.features_path, self.fnames[index]))
        return feats, os.path.splitext(self.fnames[index])[0]


def simple_collate(batch):
    """wsi_dataset.py:122-125 -> (features [B, N, D], list of slide ids)."""
    features, slide_ids = zip(*batch)
    return torch.stack(features), list(slide_ids)


class SyntheticSlideDataset(Dataset):
    """Synthetic stand-in with the same item contract (unit-normal patch features, Bernoulli stain presence with the
    ACROBAT rates of SURVEY.md section 8(d)); used by bench.py --host-input and the tests."""

    def __init__(self, n_cases, modalities: Sequence[str], n_tokens, dim, presence=(1.0, 0.46, 0.73, 0.73, 0.73), seed=0):
        self.n, self.mods, self.N, self.D = n_cases, list(modalities), n_tokens, dim
        g = torch.Generator().manual_seed(seed)
        rates = torch.tensor(list(presence)[:len(self.mods)])
        self.labels = (torch.rand(n_cases, len(self.mods), generator=g) < rates).float()
        self.labels[:, 0] = 1
        self.seed = seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        feats = [torch.randn(self.N, self.D, generator=g) if self.labels[i, m] == 1 else torch.zeros(self.N, self.D)
                 for m in range(len(self.mods))]
        return {'feats': feats, 'modality_labels': self.labels[i].tolist(), 'slide_id': f"case{i:05d}"}


class DevicePrefetcher:
    """Wraps an iterable of collate()-shaped batches: `feats` is copied into a ring of pinned host buffers and uploaded
    on a side stream `depth` batches ahead; the consumer's stream waits on the upload event only.  Yields the same
    dicts with `feats` resident on `device` (other entries untouched, labels stay on the host as the trainer expects)."""

    def __init__(self, loader: Iterable, device, depth: int = 2, drop_absent: bool = False):
        """drop_absent: the dataset fills a missing stain with an all-zero bag (wsi_dataset.py:66; ~27 % of ACROBAT's bags).
        With drop_absent only the PRESENT bags (modality_labels == 1) cross PCIe, packed [n_present, N, D]; the [B,M,N,D]
        batch is rebuilt on the device (zero fill + row scatter), bit-identical to uploading the zeros."""
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.drop_absent = bool(drop_absent)
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = []
        self._busy = {}

    def __len__(self):
        return len(self.loader)

    def _stage(self, slot, batch):
        feats = batch['feats']
        rows = None
        if self.drop_absent and 'modality_labels' in batch and feats.dim() == 4:
            present = batch['modality_labels'].reshape(-1) != 0
            if not bool(present.all()):
                rows = present.nonzero(as_tuple=True)[0]
        direct = rows is None and feats.is_pinned()        # a DataLoader(pin_memory=True) batch: uploaded from where it lies
        if not direct and (len(self._pinned) <= slot or self._pinned[slot] is None or self._pinned[slot].shape != feats.shape):
            buf = torch.empty(feats.shape, dtype=feats.dtype, pin_memory=True)   # sized for a full batch; compact uploads use a prefix
            while len(self._pinned) <= slot:
                self._pinned.append(None)
            self._pinned[slot] = buf
        pin = feats if direct else self._pinned[slot]
        if slot in self._busy:
            self._busy.pop(slot).synchronize()             # the ring slot's previous upload must have finished
        if direct:
            pass
        elif rows is None:
            pin.copy_(feats)                               # host memcpy into the pinned ring slot
        else:
            flat = feats.reshape(-1, feats.shape[2], feats.shape[3])
            pin_rows = pin.view(-1, feats.shape[2], feats.shape[3])[:rows.numel()]
            torch.index_select(flat, 0, rows, out=pin_rows)   # gather of the present bags straight into the pinned slot
        with torch.cuda.stream(self.stream):
            if rows is None:
                dev = pin.to(self.device, non_blocking=True)   # async H2D on the side stream
            else:
                up = pin_rows.to(self.device, non_blocking=True)
                dev = torch.zeros(feats.shape, dtype=feats.dtype, device=self.device)
                dev.view(-1, feats.shape[2], feats.shape[3]).index_copy_(0, rows.to(self.device, non_blocking=True), up)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._busy[slot] = ev
        out = dict(batch)
        out['feats'] = dev
        return out, ev

    def __iter__(self):
        """A daemon thread pulls batches from the loader, copies them into the pinned ring (the host memcpy releases the
        GIL) and enqueues the H2D copy on the side stream; the consumer thread only waits on the upload event, so kernel
        launches of step i overlap both the host staging and the upload of steps i+1 .. i+depth."""
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        nslots = self.depth + 2   # slots in flight: `depth` queued + 1 being consumed + 1 being filled
        stop = threading.Event()

        def worker():
            torch.cuda.set_device(self.device)
            slot = 0
            try:
                for batch in self.loader:
                    if stop.is_set():
                        return
                    item = self._stage(slot % nslots, batch)
                    slot += 1
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:  # surface loader errors in the consumer
                q.put(e)

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                batch['feats'].record_stream(cur)
                yield batch
        finally:
            stop.set()
            th.join(timeout=10.0)   # never leave the stager issuing HIP calls during interpreter teardown
