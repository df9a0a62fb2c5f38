"""Host logic of the drop-in boundary (no GPU): Hydra-style composition / overrides / interpolation and the
``config.network`` registry."""
import datetime
import os

import pytest

import mi355seg
from mi355seg.config import compose, parse_patch_size
from mi355seg.registry import build_model

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")


def test_compose_defaults_and_interpolation():
    now = datetime.datetime(2026, 1, 2, 3, 4, 5)
    c = compose(CONF, [], job_name="train", now=now)
    assert c.network == "unet" and c.in_classes == 1 and c.out_classes == 2
    assert c.output_dir == "./logs/unet"
    assert c.hydra_path.endswith(os.path.join("logs", "unet", "train-2026-01-02", "03-04-05"))
    assert c.job_name == "train" and c.init_type == "kaiming" and c.load_mode == 0
    assert parse_patch_size(c) == (64, 64, 64)


def test_overrides_and_group_selection():
    c = compose(CONF, ["config=vnet", "config.epochs=7", "config.patch_size=32", "config.init_lr=0.01", "config.ckpt=/x/y.pt"])
    assert c.network == "vnet" and c.epochs == 7 and c.init_lr == 0.01 and c.ckpt == "/x/y.pt"
    assert parse_patch_size(c) == 32
    c = compose(CONF, ["config.patch_size=16, 32, 48"])
    assert parse_patch_size(c) == (16, 32, 48)
    with pytest.raises(AssertionError):
        parse_patch_size(compose(CONF, ["config.patch_size=1,2,3,4"]))
    with pytest.raises(FileNotFoundError):
        compose(CONF, ["config=does_not_exist"])
    with pytest.raises(ValueError):
        compose(CONF, ["epochs=3"])


def test_registry_builds_in_scope_networks_with_reference_constructors():
    m = build_model({"network": "unet", "in_classes": 1, "out_classes": 2})
    assert sum(p.numel() for p in m.parameters()) == 22_581_250
    m = build_model({"network": "vnet", "in_classes": 1, "out_classes": 2})
    assert sum(p.numel() for p in m.parameters()) == 45_600_316
    m = build_model({"network": "res_unet", "in_classes": 4, "out_classes": 4})
    assert sum(p.numel() for p in m.parameters()) == 28_499_072
    m = build_model({"network": "IS", "in_classes": 1, "out_classes": 2})
    assert sum(p.numel() for p in m.parameters()) == 3 * (22_581_250 - 66) + 2 * 66      # three U-Net bodies, two heads
    m = build_model({"network": "csrnet", "in_classes": 1, "out_classes": 2})      # CSRNet(init_features=64), as the reference registry
    assert "encoder_r_1.enc1_rconv1.weight" in m.state_dict() and m.dncoder_r_3.dnc3_rconv1.weight.shape == (256, 64, 4, 4, 4)
    m = build_model({"network": "re_net", "in_classes": 1, "out_classes": 2})
    assert sum(p.numel() for p in m.parameters()) == 5_646_560
    m = build_model({"network": "er_net", "in_classes": 1, "out_classes": 2})
    assert sum(p.numel() for p in m.parameters()) == 5_110_176
    with pytest.raises(NotImplementedError):
        build_model({"network": "densenet", "in_classes": 1, "out_classes": 2})
    with pytest.raises(ValueError):
        build_model({"network": "nope", "in_classes": 1, "out_classes": 2})


def test_train_gpus_n_launches_the_repo_level_entry_point(monkeypatch):
    """``train.py config.gpus=N`` (accelerate launch of the reference, train.py:167-169): the N ranks are started on the
    repo-level ``train.py`` wrapper whatever ``sys.argv[0]`` is (pytest here), with the original overrides."""
    import sys
    from mi355seg import distributed as D
    from mi355seg import train as T
    seen = {}

    def fake_launch(nproc, argv, env=None):
        seen["nproc"], seen["argv"] = nproc, list(argv)
        return 0
    monkeypatch.setattr(D, "self_launch", fake_launch)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["/usr/bin/pytest"])
    cfg, res = T.main(["config=unet", "config.gpus=2", "config.epochs=1"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert res is None and seen["nproc"] == 2
    assert seen["argv"][0] == os.path.join(root, "train.py") and os.path.exists(seen["argv"][0])
    assert seen["argv"][1:] == ["config=unet", "config.gpus=2", "config.epochs=1"]
