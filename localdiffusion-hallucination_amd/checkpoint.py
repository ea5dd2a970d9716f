"""Checkpoint ingestion for the reference's training artefacts.

``Trainer.save`` (/root/reference/ddpm.py:1495-1507) writes ``results/<project>/model-<milestone>.pt``
as a dict  {'step', 'model', 'opt', 'ema', 'scaler'}:
  * ``data['model']``  = ``GaussianDiffusion.state_dict()``: 13 schedule buffers + ``model.<unet key>``;
  * ``data['ema']``    = ``ema_pytorch.EMA.state_dict()``: the same keys twice, under ``online_model.`` and
    ``ema_model.``, plus ``initted`` / ``step`` scalars.  Sampling uses the EMA copy
    (``trainer.ema.ema_model``, /root/reference/test.py:144-147,393).
``Trainer.load`` (:1509-1527) restores them by name -- which is why this package keeps the reference's
parameter names.  No checkpoint ships with the reference; the format is pinned by a file written by the reference's
own ``Trainer.save`` in the build container (tools/make_goldens.py g12 -> tests/golden/g12_trainer_save_manifest.json:
every key with shape and dtype), which this module reads and whose layout ``save_reference_checkpoint`` reproduces
(also checked there: the reference's ``Trainer.load`` restores a file written here).
"""
import torch

_EMA_PREFIX = "ema_model."
_ONLINE_PREFIX = "online_model."


def _read(path_or_dict, trust_pickle=False):
    """Files are read with torch's restricted unpickler (``weights_only=True``): what ``Trainer.save`` writes --
    tensors, ints, strings, plain containers -- needs nothing else (checked against a file written by the
    reference's own ``Trainer.save``, tools/make_goldens.py g12).  A file that the restricted loader rejects is NOT
    retried with the full unpickler (arbitrary code execution) unless the caller says ``trust_pickle=True``;
    I/O errors and corrupt files propagate as they are."""
    if isinstance(path_or_dict, dict):
        return path_or_dict
    if trust_pickle:
        return torch.load(path_or_dict, map_location="cpu", weights_only=False)
    import pickle
    try:
        return torch.load(path_or_dict, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        raise RuntimeError(f"{path_or_dict}: rejected by the restricted unpickler ({e}); pass trust_pickle=True "
                           "only for a file whose origin you trust") from e


def extract_state_dict(data, use_ema=True):
    """-> (state_dict for GaussianDiffusion, source tag).  Accepts the Trainer.save dict, a bare
    GaussianDiffusion state_dict, or a bare Unet state_dict."""
    if "ema" in data and use_ema and isinstance(data["ema"], dict):
        sd = {k[len(_EMA_PREFIX):]: v for k, v in data["ema"].items() if k.startswith(_EMA_PREFIX)}
        if sd:
            return sd, "ema"
        sd = {k[len(_ONLINE_PREFIX):]: v for k, v in data["ema"].items() if k.startswith(_ONLINE_PREFIX)}
        if sd:
            return sd, "ema.online"
    if "model" in data and isinstance(data["model"], dict):
        return dict(data["model"]), "model"
    return dict(data), "bare"


def load_reference_checkpoint(path_or_dict, diffusion, use_ema=True, strict=True, trust_pickle=False):
    """Load a reference checkpoint into ``diffusion`` (a ``GaussianDiffusion`` of this package, or a
    ``Unet``).  Returns {'step', 'source', 'missing', 'unexpected'}."""
    data = _read(path_or_dict, trust_pickle)
    sd, source = extract_state_dict(data, use_ema)
    own = diffusion.state_dict()
    if not any(k in own for k in sd) and any(("model." + k) in own for k in sd):
        sd = {"model." + k: v for k, v in sd.items()}                  # bare Unet dict into a GaussianDiffusion
    if not any(k in own for k in sd) and any(k.startswith("model.") and k[6:] in own for k in sd):
        sd = {k[6:]: v for k, v in sd.items() if k.startswith("model.")}   # diffusion dict into a bare Unet
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    bad = [k for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError(f"checkpoint tensors with the wrong shape: {bad[:5]}")
    if strict and (missing or unexpected):
        raise RuntimeError(f"checkpoint does not match the model: missing {missing[:5]} unexpected {unexpected[:5]}")
    keep = {k: v.to(own[k].dtype) for k, v in sd.items() if k in own}
    diffusion.load_state_dict(keep, strict=False)
    if hasattr(diffusion, "model") and hasattr(diffusion.model, "invalidate"):
        diffusion.model.invalidate()                                   # repack kernel-layout weights on next use
    step = data.get("step", None) if isinstance(data, dict) else None
    return {"step": int(step) if step is not None else None, "source": source, "missing": missing,
            "unexpected": unexpected}


def save_reference_checkpoint(diffusion, path, step=0):
    """Write ``diffusion`` in the reference's Trainer.save layout (model == ema copy; no optimizer state)."""
    sd = {k: v.detach().cpu() for k, v in diffusion.state_dict().items()}
    ema = {_ONLINE_PREFIX + k: v for k, v in sd.items()}
    ema.update({_EMA_PREFIX + k: v for k, v in sd.items()})
    ema["initted"] = torch.tensor(True)              # ema_pytorch registers both as 0-d buffers
    ema["step"] = torch.tensor(int(step))
    # 'opt': a fresh Adam state in torch's state_dict layout (no moments yet) over the model's parameters, with the
    # reference trainer's defaults (ddpm.py:1261,1265,1444): Trainer.load hands it to opt.load_state_dict (:1521)
    n_params = sum(1 for _ in diffusion.parameters())
    opt = {"state": {}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.99), "eps": 1e-8, "weight_decay": 0,
                                          "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                                          "differentiable": False, "fused": None, "params": list(range(n_params))}]}
    torch.save({"step": int(step), "model": sd, "opt": opt, "ema": ema, "scaler": None}, path)
