"""Make the reference's dotted module paths resolve to this package, so its scripts and YAML
``target:`` strings (``lvdm.modules.networks.openaimodel3d.UNetModel``, ...) pick up the MI355X
implementation with no source change:

    import t2v_turbo_amd.compat as compat; compat.install()     # before the reference imports

``install()`` puts ONE finder at the head of ``sys.meta_path``.  It answers only for the dotted names
in ``_ALIASES`` (the modules on the hot path); every other module of the reference checkout —
``utils.utils``, ``utils.common_utils``, ``lvdm.basics``, the data loaders, the reward models — keeps
importing from the checkout on ``sys.path`` exactly as before (predict.py:12-15,
train_t2v_turbo_v1_lora.py:46-69).  A parent package (``lvdm``, ``utils`` ...) is synthesised only
when nothing else on the path provides it, so the aliases also work without a checkout.
``install()`` refuses to shadow an already imported reference module unless ``force=True``."""
import importlib
import importlib.abc
import importlib.machinery
import sys

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
                            "inject_trainable_lora_extended", "monkeypatch_or_replace_lora_extended",
                            "extract_lora_ups_down", "save_lora_weight", "collapse_lora", "monkeypatch_remove_lora"]),
    "utils.lora_handler": ("lora", ["LoraHandler"]),
}
# packages that sit above an alias: taken from the checkout when it has them, synthesised otherwise
_PARENTS = {".".join(d.split(".")[:i]) for d in _ALIASES for i in range(1, len(d.split(".")))}


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    __t2v_amd_alias__ = True

    def find_spec(self, name, path=None, target=None):
        if name in _ALIASES:
            return importlib.machinery.ModuleSpec(name, self, is_package=name in _PARENTS)
        if name in _PARENTS:
            for finder in sys.meta_path:  # the real package wins: its other submodules must stay importable
                if finder is self or not hasattr(finder, "find_spec"):
                    continue
                if finder.find_spec(name, path, target) is not None:
                    return None
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        module.__t2v_amd_alias__ = True
        if module.__name__ not in _ALIASES:  # synthesised parent
            module.__path__ = []
            return
        src, names = _ALIASES[module.__name__]
        if module.__name__ in _PARENTS:
            module.__path__ = []
        mod = importlib.import_module(f"t2v_turbo_amd.{src}")
        for n in names:
            setattr(module, n, getattr(mod, n))


def installed():
    return any(getattr(f, "__t2v_amd_alias__", False) for f in sys.meta_path)


def install(force=False):
    for dotted in _ALIASES:
        m = sys.modules.get(dotted)
        if m is not None and not getattr(m, "__t2v_amd_alias__", False):
            if not force:
                raise RuntimeError(f"{dotted} is already imported from elsewhere; call compat.install() first (or force=True)")
            del sys.modules[dotted]
    if not installed():
        sys.meta_path.insert(0, _AliasFinder())
    return list(_ALIASES)


def uninstall():
    sys.meta_path[:] = [f for f in sys.meta_path if not getattr(f, "__t2v_amd_alias__", False)]
    for name in [n for n, m in sys.modules.items() if getattr(m, "__t2v_amd_alias__", False)]:
        del sys.modules[name]
