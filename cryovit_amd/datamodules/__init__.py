"""Record selection for the evaluation flow (mirror of ``/root/reference/src/cryovit/datamodules/``:
``base_datamodule.py:13-135``, ``single_sample_datamodule.py``, ``multi_sample_datamodule.py``).

The reference wraps a pandas DataFrame of ``splits.csv`` (columns ``sample, tomo_name, split_id``) in a
LightningDataModule; here the rows are plain dicts and the "dataloader" is the dataset itself with ``collate_fn`` applied
to one tomogram at a time (evaluation batch size is 1: ``configs/datamodule/dataloader/default.yaml:7``).  Only the
evaluation side (``test_df`` / ``predict_df`` and the ``val_df`` / ``train_df`` they are defined through) is mirrored."""

from __future__ import annotations

import csv
from pathlib import Path

from cryovit_amd.datasets.tomo_dataset import collate_fn


def _num(v):
    try:
        return int(float(v))
    except (TypeError, ValueError):
        return v


class BaseDataModule:
    def __init__(self, split_file, dataset_fn, dataloader_fn=None, **_):
        self.dataset_fn, self.dataloader_fn = dataset_fn, dataloader_fn
        self.split_file = Path(split_file)
        with open(self.split_file, newline="") as f:
            self.record_df = list(csv.DictReader(f))

    def _rows(self, keep, with_split: bool) -> list[dict]:
        rows = [r for r in self.record_df if keep(r)]
        return rows if with_split else [{"sample": r["sample"], "tomo_name": r["tomo_name"]} for r in rows]

    def train_df(self):
        raise NotImplementedError

    def val_df(self):
        raise NotImplementedError

    def test_df(self):
        raise NotImplementedError

    def predict_df(self):
        raise NotImplementedError

    def _loader(self, records: list[dict], what: str):
        if not records:
            raise ValueError(f"No {what} data found in the provided split file.")
        dataset = self.dataset_fn(records, train=False)
        return ((collate_fn([dataset[i]]) for i in range(len(dataset))), dataset)

    def test_dataset(self):
        records = self.test_df()
        if not records:
            raise ValueError("No testing data found in the provided split file.")
        return self.dataset_fn(records, train=False)

    def predict_dataset(self):
        records = self.predict_df()
        if not records:
            raise ValueError("No prediction data found in the provided split file.")
        return self.dataset_fn(records, train=False)


class MultiSampleDataModule(BaseDataModule):
    """Train on several samples, test on ``test_sample`` (whole samples) or else on the validation split
    (multi_sample_datamodule.py:12-101)."""

    def __init__(self, sample, split_id, split_key, test_sample=None, **kwargs):
        super().__init__(**kwargs)
        assert isinstance(sample, list), f"Multi sample 'sample' should be a list. Got {sample} instead."
        assert test_sample is None or isinstance(test_sample, list), f"Multi sample 'test_sample' should be None or a list. Got {test_sample} instead."
        self.sample, self.split_id, self.split_key, self.test_sample = sample, split_id, split_key, test_sample

    def _in_sample(self, r) -> bool:
        return r["sample"] in self.sample

    def train_df(self):
        if self.split_id is not None:
            return self._rows(lambda r: _num(r.get(self.split_key)) != self.split_id and self._in_sample(r), True)
        return self._rows(self._in_sample, False)

    def val_df(self):
        if self.split_id is None:
            return self.train_df()  # validate on the train set
        return self._rows(lambda r: _num(r.get(self.split_key)) == self.split_id and self._in_sample(r), True)

    def test_df(self):
        if self.test_sample is None:
            return self.val_df()
        return self._rows(lambda r: r["sample"] in self.test_sample, False)

    def predict_df(self):
        return self._rows(self._in_sample, False)


class SingleSampleDataModule(MultiSampleDataModule):
    """One training sample, optionally another single test sample (single_sample_datamodule.py:11-44)."""

    def __init__(self, sample, split_id, split_key, test_sample=None, **kwargs):
        assert len(sample) == 1, f"Single sample 'sample' should be a single string list. Got {sample} instead."
        assert test_sample is None or len(test_sample) == 1, f"Single sample 'test_sample' should be a single string list or None. Got {test_sample} instead."
        super().__init__(sample, split_id, split_key, test_sample, **kwargs)


__all__ = ["BaseDataModule", "SingleSampleDataModule", "MultiSampleDataModule", "collate_fn"]
