"""Checkpoint ingestion: the reference's Trainer.save layout round-trips by name."""
import pytest
import torch

import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import checkpoint, weights

CFG = dict(branch_out=True, start_intermediate=True, start_timestep=2, data="mnist", mask_x=True)
KW = dict(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist")


def _model(seed):
    net = ldh.Unet(dim=32, init_dim=32, **KW)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, seed).items()})
    return ldh.GaussianDiffusion(dict(CFG), net, image_size=28, timesteps=100, objective="pred_x0")


def test_trainer_save_layout_round_trip(tmp_path):
    src, dst = _model(3), _model(4)
    path = tmp_path / "model-best2900.pt"
    checkpoint.save_reference_checkpoint(src, str(path), step=2900)
    data = torch.load(str(path), map_location="cpu", weights_only=True)
    assert set(data) == {"step", "model", "opt", "ema", "scaler"}
    assert "ema_model.model.init_conv.weight" in data["ema"] and "online_model.betas" in data["ema"]
    info = checkpoint.load_reference_checkpoint(str(path), dst)
    assert info["step"] == 2900 and info["source"] == "ema" and not info["missing"] and not info["unexpected"]
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k


def test_ema_copy_is_preferred_and_bare_dicts_work():
    src, other, dst = _model(5), _model(6), _model(7)
    data = {"step": 7, "model": other.state_dict(),
            "ema": {"ema_model." + k: v for k, v in src.state_dict().items()}, "opt": {}, "scaler": None}
    checkpoint.load_reference_checkpoint(data, dst)
    assert torch.equal(dst.state_dict()["model.final_conv.weight"], src.state_dict()["model.final_conv.weight"])
    checkpoint.load_reference_checkpoint(data, dst, use_ema=False)
    assert torch.equal(dst.state_dict()["model.final_conv.weight"], other.state_dict()["model.final_conv.weight"])
    # a bare Unet state_dict into a GaussianDiffusion, and a diffusion dict into a bare Unet
    info = checkpoint.load_reference_checkpoint(src.model.state_dict(), dst, strict=False)
    assert info["source"] == "bare" and len(info["missing"]) == 13
    net = ldh.Unet(dim=32, init_dim=32, **KW)
    checkpoint.load_reference_checkpoint({"model": src.state_dict()}, net, strict=False)
    assert torch.equal(net.state_dict()["final_conv.bias"], src.state_dict()["model.final_conv.bias"])


def test_written_file_has_the_layout_the_reference_trainer_writes(tmp_path):
    """Manifest (every key, shape, dtype) of a file written here == manifest of the file the reference's own
    Trainer.save wrote for the same model in the build container (tools/make_goldens.py g12)."""
    import json
    import os
    import conftest
    want = json.load(open(os.path.join(conftest.GOLDEN, "g12_trainer_save_manifest.json")))

    def walk(d):
        return {k: ([list(v.shape), str(v.dtype)] if torch.is_tensor(v) else walk(v) if isinstance(v, dict) else type(v).__name__)
                for k, v in d.items()}
    path = tmp_path / "model-best2900.pt"
    checkpoint.save_reference_checkpoint(_model(3), str(path), step=2900)
    data = torch.load(str(path), map_location="cpu", weights_only=True)
    got = {k: (walk(v) if isinstance(v, dict) else type(v).__name__) for k, v in data.items()}
    got["opt"] = "dict"
    assert got == want


def test_unsafe_pickles_are_refused_unless_trusted(tmp_path):
    """A file the restricted unpickler rejects is not silently retried with the full one."""
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ("arbitrary code ran",))
    path = tmp_path / "evil.pt"
    torch.save({"model": {}, "payload": Evil()}, str(path))
    with pytest.raises(RuntimeError, match="restricted unpickler"):
        checkpoint.load_reference_checkpoint(str(path), _model(9))
    with pytest.raises(FileNotFoundError):                # I/O errors are not masked by a retry
        checkpoint.load_reference_checkpoint(str(tmp_path / "missing.pt"), _model(9))


def test_mismatch_is_an_error():
    dst = _model(8)
    wrong = ldh.Unet(dim=32, init_dim=32, mode="mri")
    with pytest.raises(RuntimeError):
        checkpoint.load_reference_checkpoint(wrong.state_dict(), dst)
