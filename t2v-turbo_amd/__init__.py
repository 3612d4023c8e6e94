"""t2v-turbo on MI355X: the VideoCrafter2 3D-UNet denoise hot path (+ KL-VAE decode) on
hand-written gfx950 HIP kernels, behind the reference's own Python interface.

Layout
  csrc/            HIP kernels + the C-ABI (include/t2v_hip.h) -> libt2v_hip.so
  native.py        ctypes binding of the C-ABI; raises if the library is missing (no fallback)
  engine.py        walks the module tree, packs weights, records the kernel sequence (hipGraph-able)
  unet3d.py        UNetModel & friends: same ctor kwargs / child names / state-dict keys as
                   lvdm/modules/networks/openaimodel3d.py + lvdm/modules/attention.py
  vae.py           AutoencoderKL / Decoder mirror of lvdm/models/autoencoder.py + ae_modules.py
  scheduler.py     T2VTurboScheduler          (scheduler/t2v_turbo_scheduler.py)
  pipeline.py      T2VTurboVC2Pipeline        (pipeline/t2v_turbo_vc2_pipeline.py)
  cd_math.py       consistency-distillation helpers + DDIMSolver (utils/common_utils.py, ode_solver/)
  lora.py          LoRA inject / collapse      (utils/lora.py)
  dist.py          flat-buffer gradient all-reduce over RCCL
  compat.py        registers the classes under the reference's dotted module paths
"""
__version__ = "0.1.0"
