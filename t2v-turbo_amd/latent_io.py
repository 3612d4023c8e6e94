"""Pre-computed latent records of the v2 training path.

On-disk format = what ``preprocess_scripts/preprocess_with_motion_prior.py:392-408`` writes per clip: a pickle of a
dict of fp16 CPU tensors ``{index, z_t, cond_teacher_out, uncond_teacher_out, score, z_example, z_example_prev,
prompt_emb}`` (optionally ``text``).  The reader mirrors ``data/mp4_dataset.py:87-154`` (``MP4LatentDataset``): a CSV with
columns ``relpath,text[,latent_root][,use_motion_guide][,short_text]``; the reference fetches ``<latent_root>/<relpath>``
from S3 with boto3 — here the same keys are resolved against a local directory (any mounted object store)."""
import csv
import os
import pickle
import random

import torch
from torch.utils.data import Dataset

RECORD_KEYS = ("index", "z_t", "cond_teacher_out", "uncond_teacher_out", "score", "z_example", "z_example_prev", "prompt_emb")


def pack_latent_record(index, z_t, cond_teacher_out, uncond_teacher_out, score, z_example, z_example_prev, prompt_emb,
                       text=None):
    """bytes of one record (preprocess_with_motion_prior.py:392-403): every tensor fp16, detached, on CPU."""
    rec = {"index": index}
    for k, v in (("z_t", z_t), ("cond_teacher_out", cond_teacher_out), ("uncond_teacher_out", uncond_teacher_out),
                 ("score", score), ("z_example", z_example), ("z_example_prev", z_example_prev), ("prompt_emb", prompt_emb)):
        rec[k] = v.to(torch.float16)
    rec = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in rec.items()}
    if text is not None:
        rec["text"] = text
    return pickle.dumps(rec)


def unpack_latent_record(blob):
    rec = pickle.loads(blob)
    missing = [k for k in RECORD_KEYS if k not in rec]
    if missing:
        raise KeyError(f"latent record lacks {missing}")
    return rec


def _parse_bool(v):
    """The spellings ``pd.read_csv`` accepts / writes for a boolean column: 1, 1.0, True, true (anything else is False);
    raises on text that is none of the known true / false spellings."""
    t = str(v).strip().lower()
    if t in ("1", "1.0", "true", "t", "yes"):
        return True
    if t in ("0", "0.0", "false", "f", "no"):
        return False
    raise ValueError(f"use_motion_guide: cannot read {v!r} as a boolean")


class LatentRecordDataset(Dataset):
    """``MP4LatentDataset`` (data/mp4_dataset.py:87-154) over a local root instead of an S3 bucket."""
    MAX_RETRIES = 16

    def __init__(self, path_to_csv, latent_root="latent_root", root_dir="."):
        self.latent_root, self.root_dir = latent_root, root_dir
        with open(path_to_csv, newline="") as f:
            reader = csv.DictReader(f)
            self.rows = list(reader)
            fields = reader.fieldnames or []
        self.length = len(self.rows)
        # CSV schema / parse errors are the annotation file's, not a record's: found here, once, for every row — resampling
        # on them would spin forever.  Everything raised later, while a record file is read or checked, is a per-record fault.
        # A missing COLUMN is the file's fault (raise here); a short ROW (DictReader yields None for its missing cells) is one bad
        # record: the reference tolerates those (data/mp4_dataset.py: pandas fills NaN, the record is resampled on the assert) and so
        # does __getitem__'s bounded resampling.
        for col in ("relpath", "text"):
            if col not in fields:
                raise KeyError(f"{path_to_csv}: lacks the column {col!r}")
        for i, row in enumerate(self.rows):
            if row.get("use_motion_guide") not in (None, ""):
                row["use_motion_guide"] = _parse_bool(row["use_motion_guide"])

    def __len__(self):
        return self.length

    def get_latent_text_pair(self, idx):
        row = self.rows[idx]
        relpath, text = row["relpath"], row["text"]
        if relpath is None or text is None:
            raise ValueError(f"annotation row {idx} is short (no relpath / text)")   # a per-record fault: resampled by __getitem__
        root = row.get("latent_root") or self.latent_root
        latent_dir = f"{root}/{relpath}"
        use_motion_guide = row["use_motion_guide"] if row.get("use_motion_guide") not in (None, "") else True   # parsed in __init__
        short_text = row.get("short_text") or ""
        if str(short_text) == "nan":
            short_text = ""
        with open(os.path.join(self.root_dir, latent_dir), "rb") as f:
            latent_dict = pickle.loads(f.read())
        if "webvid" in latent_dir:
            text = latent_dict.pop("text")
            short_text = text
        elif "text" in latent_dict:
            assert text == latent_dict.pop("text")
        return latent_dict, text, short_text, use_motion_guide

    def __getitem__(self, idx):
        tries = 0
        while True:  # ANY fault of a record (missing file, damaged pickle, a webvid record without text, a text mismatch) is
            # answered by a random other record, as the reference does (data/mp4_dataset.py:139-154); bounded here
            try:
                latent_dict, text, short_text, use_motion_guide = self.get_latent_text_pair(idx)
                for k in latent_dict:
                    if isinstance(latent_dict[k], torch.Tensor):
                        latent_dict[k] = latent_dict[k].detach().cpu()
                sample = dict(txt=text, short_txt=short_text, use_motion_guide=use_motion_guide)
                sample.update(latent_dict)
                return sample
            except Exception:
                tries += 1
                if self.length <= 1 or tries >= self.MAX_RETRIES:
                    raise
                idx = random.randint(0, self.length - 1)
