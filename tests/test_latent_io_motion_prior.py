"""Latent record format + motion-prior helpers, checked against the restated reference
arithmetic: record keys / dtypes of preprocess_with_motion_prior.py:392-403, DDIM inversion algebra of
ode_solver/ddim_solver.py:89-97 (inverting a deterministic DDIM step), rank-1 motion loss of common_utils.py:446-478."""
import csv
import os
import pickle

import pytest
import torch

from t2v_turbo_amd import cd_math, latent_io, motion_prior
from t2v_turbo_amd.scheduler import T2VTurboScheduler
from tests.util import tiny_unet_params


def _record(i=3):
    g = torch.Generator().manual_seed(i)
    lat = lambda: torch.randn(4, 8, 4, 4, generator=g)
    return dict(index=i, z_t=lat(), cond_teacher_out=lat(), uncond_teacher_out=lat(), score=lat(), z_example=lat(),
                z_example_prev=lat(), prompt_emb=torch.randn(77, 32, generator=g))


def test_record_round_trip_and_dataset(tmp_path):
    rec = _record()
    blob = latent_io.pack_latent_record(**rec, text="a cat")
    raw = pickle.loads(blob)
    assert set(raw) == set(latent_io.RECORD_KEYS) | {"text"} and raw["index"] == 3
    for k in latent_io.RECORD_KEYS[1:]:
        assert raw[k].dtype == torch.float16 and raw[k].device.type == "cpu"
        assert torch.equal(raw[k], rec[k].half())
    back = latent_io.unpack_latent_record(blob)
    assert torch.equal(back["score"], rec["score"].half())
    with pytest.raises(KeyError):
        latent_io.unpack_latent_record(pickle.dumps({"index": 0}))
    os.makedirs(tmp_path / "lat" / "clips")
    for name, txt in (("clips/a.pkl", "a cat"), ("clips/b.pkl", "a dog")):
        with open(tmp_path / "lat" / name, "wb") as f:
            f.write(latent_io.pack_latent_record(**_record(), text=txt))
    with open(tmp_path / "meta.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["relpath", "text", "use_motion_guide", "short_text"])
        w.writerow(["clips/a.pkl", "a cat", "0", "cat"])
        w.writerow(["clips/b.pkl", "a dog", "", ""])
    ds = latent_io.LatentRecordDataset(str(tmp_path / "meta.csv"), latent_root="lat", root_dir=str(tmp_path))
    assert len(ds) == 2
    s0, s1 = ds[0], ds[1]
    assert s0["txt"] == "a cat" and s0["short_txt"] == "cat" and s0["use_motion_guide"] is False and "text" not in s0
    assert s1["use_motion_guide"] is True and s1["short_txt"] == "" and s1["z_t"].dtype == torch.float16


def test_ddim_inversion_inverts_the_ddim_step():
    """x_t = reverse_step(x_prev, eps, t) must be undone by the forward DDIM step with the same eps."""
    sched = T2VTurboScheduler(linear_start=0.00085, linear_end=0.012)
    solver = cd_math.DDIMSolver(sched.alphas_cumprod.numpy(), ddim_timesteps=50)
    x0 = torch.randn(1, 4, 4, 8, 8, dtype=torch.float64, generator=torch.Generator().manual_seed(0))

    class Eps(torch.nn.Module):  # a fixed "noise prediction" so both directions use the same eps
        def forward(self, x, ts, **kw):
            return torch.tanh(x.flip(-1)) * 0.5

    lat = motion_prior.reverse_ddim_loop(x0, Eps(), {}, solver, 3, torch.device("cpu"))
    assert len(lat) == 3 and lat[0].shape == x0.shape
    ts0 = solver.ddim_timesteps[torch.tensor([0])].long()
    eps = torch.tanh(x0.flip(-1)) * 0.5
    a_t = solver.alpha_cumprods[ts0].double()
    prev = (ts0 - solver.step_ratio).clip(min=0)
    a_prev = solver.alpha_cumprods[prev].double()
    x_back = a_prev.sqrt() * (lat[0] - (1 - a_t).sqrt() * eps) / a_t.sqrt() + (1 - a_prev).sqrt() * eps
    # the solver's coefficient tables are fp32 (as the reference's): sqrt(a_next / a) rounded in fp32 against sqrt(a_next) / sqrt(a)
    # in fp64 differ by ~6e-8 relative — the identity holds to that, not to fp64 round-off (an atol of 1e-9 passed or failed with
    # the draw of x0)
    assert torch.allclose(x_back, x0, atol=2e-6)


def test_motion_rank_loss_and_score():
    ref = torch.tensor([[[0.1, 0.7, 0.2], [0.5, 0.3, 0.2]]])
    gen = torch.tensor([[[0.3, 0.4, 0.3], [0.2, 0.3, 0.5]]])
    want = ((0.7 - 0.4) ** 2 + (0.5 - 0.2) ** 2) / 2
    assert abs(float(motion_prior.calculate_motion_rank_new(ref, gen, 1)) - want) < 1e-7
    assert float(motion_prior.calculate_motion_rank_new(ref, gen, 0)) == 0.0
    with pytest.raises(ValueError):
        motion_prior.calculate_motion_rank_new(ref, gen, 4)
    assert abs(float(motion_prior.compute_temp_loss({"a": gen, "b": gen}, {"a": ref, "b": ref})) - 100 * want) < 1e-5
    # end to end on the tiny UNet mirror (torch path, CPU): score has the latent's shape and is non-zero
    from oracle import synth
    from t2v_turbo_amd.unet3d import UNetModel
    p = tiny_unet_params(record_attn_probs=True)
    unet = UNetModel(**p).eval()
    unet.load_state_dict(synth.synth_state_dict(synth.manifest_of(unet)))
    x, ctx = synth.synth_inputs(1, 4, 8, 8, ctx_len=7, ctx_dim=p["context_dim"])
    ts = torch.tensor([519])
    context = {"context": ctx, "fps": 16, "timestep_cond": torch.randn(1, 256, generator=torch.Generator().manual_seed(1))}
    score, out = motion_prior.get_motion_prior_score(unet, x.clone(), ts, x * 0.9, context, context, 10.0)
    assert score.shape == x.shape and out.shape == x.shape and float(score.abs().sum()) > 0


def test_dataset_reads_the_boolean_spellings_pandas_writes_and_fails_loudly_on_bad_ones(tmp_path):
    """``use_motion_guide`` as pd.read_csv accepts it (data/mp4_dataset.py:87-154 reads the CSV with pandas): True / False / 1.0 /
    0 / empty; a value that is no boolean is a schema error and must propagate instead of being resampled away forever."""
    os.makedirs(tmp_path / "lat")
    with open(tmp_path / "lat" / "a.pkl", "wb") as f:
        f.write(latent_io.pack_latent_record(**_record(), text="a cat"))
    rows = [("True", True), ("False", False), ("1.0", True), ("0", False), ("", True), ("true", True)]

    def write(extra):
        with open(tmp_path / "meta.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["relpath", "text", "use_motion_guide", "short_text"])
            for v, _ in rows:
                w.writerow(["a.pkl", "a cat", v, "cat"])
            for r in extra:
                w.writerow(r)

    write([])
    ds = latent_io.LatentRecordDataset(str(tmp_path / "meta.csv"), latent_root="lat", root_dir=str(tmp_path))
    for i, (_, want) in enumerate(rows):
        assert ds[i]["use_motion_guide"] is want
    # schema / parse errors belong to the annotation file: raised when it is opened, for any row, never resampled away
    write([["a.pkl", "a cat", "maybe", "cat"]])
    with pytest.raises(ValueError):
        latent_io.LatentRecordDataset(str(tmp_path / "meta.csv"), latent_root="lat", root_dir=str(tmp_path))


def test_dataset_resamples_on_any_per_record_fault(tmp_path):
    """data/mp4_dataset.py:139-154 answers ANY exception of a record with a random other record: a missing file, a damaged
    pickle, a text that does not match the CSV, a webvid record without text must not abort a long training run."""
    import random
    os.makedirs(tmp_path / "lat" / "webvid")
    with open(tmp_path / "lat" / "good.pkl", "wb") as f:
        f.write(latent_io.pack_latent_record(**_record(), text="a cat"))
    with open(tmp_path / "lat" / "mismatch.pkl", "wb") as f:
        f.write(latent_io.pack_latent_record(**_record(), text="a dog"))
    with open(tmp_path / "lat" / "damaged.pkl", "wb") as f:
        f.write(b"\x80\x04not a pickle")
    with open(tmp_path / "lat" / "webvid" / "notext.pkl", "wb") as f:
        f.write(latent_io.pack_latent_record(**_record()))
    with open(tmp_path / "meta.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["relpath", "text"])
        for rel in ("good.pkl", "mismatch.pkl", "damaged.pkl", "webvid/notext.pkl", "missing.pkl"):
            w.writerow([rel, "a cat"])
    ds = latent_io.LatentRecordDataset(str(tmp_path / "meta.csv"), latent_root="lat", root_dir=str(tmp_path))
    random.seed(0)
    for i in range(5):
        assert ds[i]["txt"] == "a cat"      # every faulty row ends up on the one good record


def test_short_csv_row_is_a_record_fault_not_a_constructor_error(tmp_path):
    """A row with fewer cells than the header (DictReader yields None) is ONE bad record: the constructor accepts the file and the
    record itself is refused when read (``__getitem__`` resamples on that, as data/mp4_dataset.py:139-154 does on any per-record
    fault); a missing COLUMN still raises at construction."""
    import csv as _csv

    import pytest as _pytest

    from t2v_turbo_amd.latent_io import LatentRecordDataset

    p = tmp_path / "ann.csv"
    with open(p, "w", newline="") as f:
        w = _csv.writer(f)
        w.writerow(["relpath", "text"])
        w.writerow(["a.mp4", "a cat"])
        w.writerow(["b.mp4"])                       # short row
    ds = LatentRecordDataset(str(p), latent_root="lat", root_dir=str(tmp_path))
    assert len(ds) == 2
    with _pytest.raises(ValueError, match="short"):
        ds.get_latent_text_pair(1)
    bad = tmp_path / "bad.csv"
    with open(bad, "w", newline="") as f:
        w = _csv.writer(f)
        w.writerow(["relpath"])
        w.writerow(["a.mp4"])
    with _pytest.raises(KeyError):
        LatentRecordDataset(str(bad), latent_root="lat", root_dir=str(tmp_path))
