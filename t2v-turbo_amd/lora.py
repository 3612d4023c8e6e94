"""LoRA for the UNet leaves, interface-compatible with the reference's ``utils/lora.py``
(cloneofsimo-style): ``LoraInjected{Linear,Conv2d,Conv3d}`` with ``.linear|.conv``,
``.lora_down``, ``.lora_up``, ``.dropout``, ``.scale``, ``.selector``; injection walks every module
whose class NAME is in ``target_replace_module`` and replaces descendants whose class is EXACTLY
``nn.Linear`` / ``nn.Conv2d`` / ``nn.Conv3d`` in ``named_modules()`` order, so the flat
``[up0, down0, up1, down1, ...]`` list of ``unet_lora.pt`` lines up with the reference's
(utils/lora.py:263-307,387-486,582-594).

Inference never pays for the branch: the native engine merges ``W + scale * up @ down`` when it packs
weights (engine.effective_weight_bias), and ``collapse_lora`` bakes it into the base weight."""
import torch
import torch.nn as nn

UNET_EXTENDED_TARGET_REPLACE = {"UNetModel"}


class _LoraInjected(nn.Module):
    def forward(self, x):
        base = getattr(self, "linear", None) or self.conv
        return base(x) + self.dropout(self.lora_up(self.selector(self.lora_down(x)))) * self.scale

    def realize_as_lora(self):
        return self.lora_up.weight.data * self.scale, self.lora_down.weight.data

    def _init(self, r):
        nn.init.normal_(self.lora_down.weight, std=1 / r)
        nn.init.zeros_(self.lora_up.weight)


def _clamp_rank(r, a, b):
    if r > min(a, b):
        print(f"LoRA rank {r} is too large. setting to: {min(a, b)}")
        r = min(a, b)
    return r


class LoraInjectedLinear(_LoraInjected):
    def __init__(self, in_features, out_features, bias=False, r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        r = _clamp_rank(r, in_features, out_features)
        self.r = r
        self.linear = nn.Linear(in_features, out_features, bias)
        self.lora_down = nn.Linear(in_features, r, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Linear(r, out_features, bias=False)
        self.scale = scale
        self.selector = nn.Identity()
        self._init(r)

    def set_selector_from_diag(self, diag):
        assert diag.shape == (self.r,)
        self.selector = nn.Linear(self.r, self.r, bias=False)
        self.selector.weight.data = torch.diag(diag).to(self.lora_up.weight.device, self.lora_up.weight.dtype)


class LoraInjectedConv2d(_LoraInjected):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        r = _clamp_rank(r, in_channels, out_channels)
        self.r = r
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.lora_down = nn.Conv2d(in_channels, r, kernel_size, stride, padding, dilation, groups, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv2d(r, out_channels, 1, 1, 0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        self._init(r)

    def set_selector_from_diag(self, diag):
        assert diag.shape == (self.r,)
        self.selector = nn.Conv2d(self.r, self.r, 1, 1, 0, bias=False)
        self.selector.weight.data = torch.diag(diag)[:, :, None, None].to(self.lora_up.weight.device, self.lora_up.weight.dtype)


class LoraInjectedConv3d(_LoraInjected):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 1, 1), padding=(1, 0, 0), bias=False, r=4,
                 dropout_p=0, scale=1.0):
        super().__init__()
        r = _clamp_rank(r, in_channels, out_channels)
        self.r = r
        self.kernel_size, self.padding = kernel_size, padding
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, padding=padding, bias=bias)
        self.lora_down = nn.Conv3d(in_channels, r, kernel_size=kernel_size, bias=False, padding=padding)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv3d(r, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        self._init(r)

    def set_selector_from_diag(self, diag):
        assert diag.shape == (self.r,)
        self.selector = nn.Conv3d(self.r, self.r, 1, 1, 0, bias=False)
        self.selector.weight.data = torch.diag(diag)[:, :, None, None, None].to(self.lora_up.weight.device,
                                                                                  self.lora_up.weight.dtype)


_INJECTED = (LoraInjectedLinear, LoraInjectedConv2d, LoraInjectedConv3d)


def find_modules(model, ancestor_class=None, search_class=(nn.Linear,), exclude_children_of=_INJECTED):
    """Yield (parent, child_name, child) for descendants of the named ancestors (utils/lora.py:263-307)."""
    if ancestor_class is None:  # every module once (the reference re-visits nested ancestors; same set)
        ancestors = [model]
    else:
        ancestors = [m for m in model.modules() if m.__class__.__name__ in ancestor_class]
    for ancestor in ancestors:
        for fullname, module in list(ancestor.named_modules()):
            if not isinstance(module, tuple(search_class)) or fullname == "":
                continue
            *path, name = fullname.split(".")
            parent = ancestor
            for part in path:
                parent = parent.get_submodule(part)
            if exclude_children_of and isinstance(parent, tuple(exclude_children_of)):
                continue
            if parent._modules.get(name) is not module:  # already swapped while iterating
                continue
            yield parent, name, module


def inject_trainable_lora_extended(model, target_replace_module=UNET_EXTENDED_TARGET_REPLACE, r=4, loras=None):
    """Replace exact-class Linear/Conv2d/Conv3d leaves by their LoRA-injected form; returns
    (list of parameter iterators [up, down, up, down, ...], list of child names)."""
    params, names = [], []
    if loras is not None:
        loras = torch.load(loras)
    for parent, name, child in find_modules(model, target_replace_module, (nn.Linear, nn.Conv2d, nn.Conv3d)):
        cls = child.__class__
        if cls is nn.Linear:
            new = LoraInjectedLinear(child.in_features, child.out_features, child.bias is not None, r=r)
            new.linear.weight = child.weight
            if child.bias is not None:
                new.linear.bias = child.bias
        elif cls is nn.Conv2d:
            new = LoraInjectedConv2d(child.in_channels, child.out_channels, child.kernel_size, child.stride,
                                     child.padding, child.dilation, child.groups, child.bias is not None, r=r)
            new.conv.weight = child.weight
            if child.bias is not None:
                new.conv.bias = child.bias
        elif cls is nn.Conv3d:
            new = LoraInjectedConv3d(child.in_channels, child.out_channels, bias=child.bias is not None,
                                     kernel_size=child.kernel_size, padding=child.padding, r=r)
            new.conv.weight = child.weight
            if child.bias is not None:
                new.conv.bias = child.bias
        else:
            continue
        new.to(child.weight.device).to(child.weight.dtype)
        parent._modules[name] = new
        if loras is not None:
            up, down = loras.pop(0), loras.pop(0)
            new.lora_up.weight = up if isinstance(up, nn.Parameter) else nn.Parameter(up)
            new.lora_down.weight = down if isinstance(down, nn.Parameter) else nn.Parameter(down)
        new.lora_up.weight.requires_grad = True
        new.lora_down.weight.requires_grad = True
        params.append(new.lora_up.parameters())
        params.append(new.lora_down.parameters())
        names.append(name)
    return params, names


def extract_lora_ups_down(model, target_replace_module=UNET_EXTENDED_TARGET_REPLACE):
    out = [(c.lora_up, c.lora_down) for _, _, c in find_modules(model, target_replace_module, _INJECTED, None)]
    if not out:
        raise ValueError("No lora injected.")
    return out


def save_lora_weight(model, path="./lora.pt", target_replace_module=UNET_EXTENDED_TARGET_REPLACE):
    weights = []
    for up, down in extract_lora_ups_down(model, target_replace_module):
        weights.append(up.weight.to("cpu").to(torch.float32))
        weights.append(down.weight.to("cpu").to(torch.float32))
    torch.save(weights, path)


def collapse_lora(model, replace_modules=UNET_EXTENDED_TARGET_REPLACE, alpha=1.0):
    """W <- W + alpha * up @ down on every injected leaf (utils/lora.py:793-830); the branch then
    contributes on top unless it is removed (``monkeypatch_remove_lora``), as in the reference."""
    for _, _, c in find_modules(model, replace_modules, _INJECTED, None):
        base = getattr(c, "linear", None) or c.conv
        delta = c.lora_up.weight.data.flatten(1) @ c.lora_down.weight.data.flatten(1)
        base.weight = nn.Parameter(base.weight.data + alpha * delta.reshape(base.weight.shape).type(base.weight.dtype)
                                   .to(base.weight.device))


def monkeypatch_remove_lora(model):
    """Put the plain leaf (sharing the base weight tensors) back (utils/lora.py:1013-1062)."""
    for parent, name, c in find_modules(model, None, _INJECTED, None):
        src = getattr(c, "linear", None) or c.conv
        if isinstance(src, nn.Linear):
            new = nn.Linear(src.in_features, src.out_features, src.bias is not None)
        elif isinstance(src, nn.Conv2d):
            new = nn.Conv2d(src.in_channels, src.out_channels, src.kernel_size, src.stride, src.padding, src.dilation,
                            src.groups, src.bias is not None)
        else:
            new = nn.Conv3d(src.in_channels, src.out_channels, kernel_size=src.kernel_size, padding=src.padding,
                            bias=src.bias is not None)
        new.weight = src.weight
        if src.bias is not None:
            new.bias = src.bias
        parent._modules[name] = new


def monkeypatch_or_replace_lora_extended(model, loras, target_replace_module=UNET_EXTENDED_TARGET_REPLACE, r=4):
    """Load a flat ``[up0, down0, up1, down1, ...]`` list into ``model`` (utils/lora.py:877-997): plain leaves
    are injected, already injected leaves get fresh branches of rank ``r`` (an int, or a list consumed
    per leaf); a leaf whose kind does not match the rank of the next tensor is skipped, as in the reference."""
    kinds = {nn.Linear: 2, LoraInjectedLinear: 2, nn.Conv2d: 4, LoraInjectedConv2d: 4, nn.Conv3d: 5, LoraInjectedConv3d: 5}
    for parent, name, child in find_modules(model, target_replace_module, tuple(kinds)):
        if child.__class__ not in kinds or not loras or loras[0].dim() != kinds[child.__class__]:
            continue
        src = child if not isinstance(child, _INJECTED) else (getattr(child, "linear", None) or child.conv)
        rank = r.pop(0) if isinstance(r, list) else r
        if isinstance(src, nn.Linear):
            new = LoraInjectedLinear(src.in_features, src.out_features, src.bias is not None, r=rank)
            new.linear.weight, new.linear.bias = src.weight, src.bias
        elif isinstance(src, nn.Conv2d):
            new = LoraInjectedConv2d(src.in_channels, src.out_channels, src.kernel_size, src.stride, src.padding,
                                     src.dilation, src.groups, src.bias is not None, r=rank)
            new.conv.weight, new.conv.bias = src.weight, src.bias
        else:
            new = LoraInjectedConv3d(src.in_channels, src.out_channels, bias=src.bias is not None,
                                     kernel_size=src.kernel_size, padding=src.padding, r=rank)
            new.conv.weight, new.conv.bias = src.weight, src.bias
        up, down = loras.pop(0), loras.pop(0)
        new.lora_up.weight = nn.Parameter(up.type(src.weight.dtype))
        new.lora_down.weight = nn.Parameter(down.type(src.weight.dtype))
        parent._modules[name] = new.to(src.weight.device)


class LoraHandler:
    """The trainers' and demos' entry point to the injection (utils/lora_handler.py:46-160;
    train_t2v_turbo_v1_lora.py:644-657, app.py:249-263).  Only the ``cloneofsimo`` flavour exists upstream.
    As upstream, ``dropout`` is accepted and NOT forwarded to the injector (lora_handler.py:84-100 filters the
    arguments to model / loras / target_replace_module / r), so injected leaves keep the default p = 0.1."""

    def __init__(self, version="cloneofsimo", use_unet_lora=False, use_text_lora=False, save_for_webui=False,
                 only_for_webui=False, lora_bias="none", unet_replace_modules=("UNet3DConditionModel",)):
        assert version == "cloneofsimo"
        self.version = version
        self.lora_loader = monkeypatch_or_replace_lora_extended
        self.lora_injector = inject_trainable_lora_extended
        self.lora_bias = lora_bias
        self.use_unet_lora = use_unet_lora
        self.use_text_lora = use_text_lora
        self.save_for_webui = save_for_webui
        self.only_for_webui = only_for_webui
        self.unet_replace_modules = list(unet_replace_modules)
        self.use_lora = any([use_text_lora, use_unet_lora])

    def add_lora_to_model(self, use_lora, model, replace_modules, dropout=0.0, lora_path=None, r=16):
        """Returns (list of parameter iterators, child names); (model, None) when ``use_lora`` is false."""
        if not use_lora:
            return model, None
        params, negation = self.lora_injector(model=model, loras=lora_path, target_replace_module=replace_modules, r=r)
        extract_lora_ups_down(model, target_replace_module=replace_modules)  # raises if nothing was injected
        return params, negation


def lora_parameters(model):
    """Trainable LoRA tensors in injection order (the buffer DDP all-reduces every step)."""
    out = []
    for up, down in extract_lora_ups_down(model, None):
        out += [up.weight, down.weight]
    return out
