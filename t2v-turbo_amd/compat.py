"""Make the reference's dotted module paths resolve to this package, so its scripts and YAML
``target:`` strings (``lvdm.modules.networks.openaimodel3d.UNetModel``, ...) pick up the MI355X
implementation with no source change:

    import t2v_turbo_amd.compat as compat; compat.install()     # before the reference imports

``install()`` registers alias modules in ``sys.modules``; it refuses to shadow an already imported
reference module unless ``force=True``."""
import sys
import types

_ALIASES = {
    "lvdm.modules.networks.openaimodel3d": ("unet3d", ["UNetModel", "ResBlock", "TemporalConvBlock", "Downsample",
                                                        "Upsample", "TimestepEmbedSequential", "TimestepBlock"]),
    "lvdm.modules.attention": ("unet3d", ["SpatialTransformer", "TemporalTransformer", "BasicTransformerBlock",
                                          "CrossAttention", "FeedForward", "GEGLU"]),
    "lvdm.modules.networks.ae_modules": ("vae", ["Encoder", "Decoder", "ResnetBlock", "AttnBlock", "Upsample",
                                                 "Downsample", "Normalize", "make_attn"]),
    "lvdm.models.autoencoder": ("vae", ["AutoencoderKL"]),
    "lvdm.distributions": ("vae", ["DiagonalGaussianDistribution"]),
    "lvdm.models.ddpm3d": ("latent_diffusion", ["LatentDiffusion", "DiffusionWrapper"]),
    "scheduler.t2v_turbo_scheduler": ("scheduler", ["T2VTurboScheduler", "T2VTurboSchedulerOutput"]),
    "pipeline.t2v_turbo_vc2_pipeline": ("pipeline", ["T2VTurboVC2Pipeline"]),
    "ode_solver.ddim_solver": ("cd_math", ["DDIMSolver"]),
    "ode_solver": ("cd_math", ["DDIMSolver"]),
    "model_scope.unet_3d_condition": ("ms_unet3d", ["UNet3DConditionModel", "UNet3DConditionOutput"]),
    "model_scope.unet_3d_blocks": ("ms_unet3d", ["CrossAttnDownBlock3D", "DownBlock3D", "UNetMidBlock3DCrossAttn",
                                                 "CrossAttnUpBlock3D", "UpBlock3D"]),
    "utils.lora": ("lora", ["LoraInjectedLinear", "LoraInjectedConv2d", "LoraInjectedConv3d",
                            "inject_trainable_lora_extended", "extract_lora_ups_down", "save_lora_weight",
                            "collapse_lora", "monkeypatch_remove_lora"]),
}


def install(force=False):
    import importlib
    made = []
    for dotted, (src, names) in _ALIASES.items():
        if dotted in sys.modules and not force and not getattr(sys.modules[dotted], "__t2v_amd_alias__", False):
            raise RuntimeError(f"{dotted} is already imported from elsewhere; call compat.install() first (or force=True)")
        mod = importlib.import_module(f"t2v_turbo_amd.{src}")
        parts = dotted.split(".")
        for i in range(1, len(parts)):  # parent packages
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                m.__t2v_amd_alias__ = True
                sys.modules[pkg] = m
        alias = types.ModuleType(dotted)
        alias.__t2v_amd_alias__ = True
        for n in names:
            setattr(alias, n, getattr(mod, n))
        sys.modules[dotted] = alias
        parent = sys.modules.get(".".join(parts[:-1]))
        if parent is not None:
            setattr(parent, parts[-1], alias)
        made.append(dotted)
    return made
