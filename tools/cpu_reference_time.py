#!/usr/bin/env python
"""CPU baseline as SURVEY.md 8(d) / BASELINE.md 3 specify it: the IMPORTED reference ``UNetModel`` (fp32, eval,
use_checkpoint=False, vanilla attention, no_grad) on the bench's synthetic inputs, 1 warm-up + 3 runs, median — next to the
oracle (``oracle.unet_oracle``, the CPU restatement bench.py times on the GPU box, where /root/reference does not exist) on
the same weights, inputs and threads.  Runs where the reference checkout is (this container), writes one JSON file.

    python tools/cpu_reference_time.py --frames 16 --out profiles/r02_cpu_reference_timing.json
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference_timing.json"))
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    mg.install_stubs()
    from lvdm.modules.networks.openaimodel3d import UNetModel as RefUNet
    import bench
    from oracle import unet_oracle as uo
    from t2v_turbo_amd.nn_util import guidance_embedding

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    cfg = mg.full_unet_params()
    assert {k: cfg[k] for k in bench.VC2_UNET if k in cfg} == {k: (list(v) if isinstance(v, tuple) else v)
                                                              for k, v in bench.VC2_UNET.items() if k in cfg}, "bench config differs from the yaml"
    torch.manual_seed(1234)
    ref = RefUNet(**cfg).eval()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for p in ref.parameters():  # zero_module'd tensors re-drawn, as bench.build_model does
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    sd = {k: v.detach().float() for k, v in ref.state_dict().items()}
    gi = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 16, 40, 64, generator=gi)[:, :, :a.frames].contiguous()
    ctx = torch.randn(1, 77, 1024, generator=gi)
    tc = guidance_embedding(torch.tensor([7.5]), 256)
    ts = torch.tensor([999])

    def run_ref():
        with torch.no_grad():
            return ref(x, ts, context=ctx, fps=16, timestep_cond=tc)

    def run_oracle():
        return uo.unet_forward(sd, bench.VC2_UNET, x, ts, ctx, fps=16, timestep_cond=tc)

    out = {"frames": a.frames, "threads": threads, "cpu": platform.processor() or platform.machine(), "runs": a.runs}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    res = {}
    for name, fn in (("reference", run_ref), ("oracle", run_oracle)):
        t0 = time.time()
        y = fn()
        warm = time.time() - t0
        times = []
        for _ in range(a.runs):
            t0 = time.time()
            y = fn()
            times.append(time.time() - t0)
        res[name] = y
        out[name] = {"warmup_s": round(warm, 2), "runs_s": [round(t, 2) for t in times], "median_s": round(statistics.median(times), 2),
                     "steps_per_s_16f_equiv": round((a.frames / 16.0) / statistics.median(times), 5)}
        print(name, out[name], flush=True)
    d = (res["oracle"] - res["reference"]).double().norm() / res["reference"].double().norm()
    out["oracle_vs_reference_rel_l2"] = float(d)
    out["oracle_over_reference_time"] = round(out["oracle"]["median_s"] / out["reference"]["median_s"], 3)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
