"""CLI surface and the cv2/torchvision-free input pipeline (reference train.py:30-66, test.py:28-58, utils.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-transformer_amd")

TRAIN_FLAGS = ["-j", "--workers", "--epochs", "--start-epoch", "-b", "--batch_size", "--learning_rate", "--momentum",
               "--weight-decay", "--wd", "--train_dataset", "--val_dataset", "--save_freq", "--modelname", "--cuda",
               "--aug", "--load", "--save", "--direc", "--crop", "--imgsize", "--device", "--gray"]
TEST_FLAGS = ["-j", "--workers", "--epochs", "--start-epoch", "-b", "--batch_size", "--learning_rate", "--momentum",
              "--weight-decay", "--wd", "--train_dataset", "--val_dataset", "--save_freq", "--modelname", "--cuda",
              "--direc", "--crop", "--device", "--loaddirec", "--imgsize", "--gray"]


def _flags(script):
    import importlib.util
    spec = importlib.util.spec_from_file_location("cli_" + script, os.path.join(PKG, script + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {s for a in mod.parser._actions for s in a.option_strings}, mod.parser


def test_train_flags_and_defaults():
    flags, parser = _flags("train")
    assert set(TRAIN_FLAGS) <= flags
    a = parser.parse_args(["--train_dataset", "x", "--epoch", "3"])       # README.md:113 relies on prefix matching
    assert (a.epochs, a.batch_size, a.learning_rate, a.save_freq, a.modelname, a.direc, a.gray) == (3, 1, 1e-3, 10, "MedT", "./medt", "no")


def test_test_flags():
    flags, parser = _flags("test")
    assert set(TEST_FLAGS) <= flags
    a = parser.parse_args(["--loaddirec", "m.pth", "--val_dataset", "v", "--modelname", "gatedaxialunet", "--imgsize", "128"])
    assert a.save_freq == 5 and a.direc == "./results"


def test_dataset_pipeline(tmp_path):
    from medt_amd.data import make_synthetic_dataset, imread
    import utils
    import utils_gray
    root = make_synthetic_dataset(str(tmp_path / "d"), n=4, size=32, seed=1)
    ds = utils.ImageToImage2D(root, utils.JointTransform2D(crop=None, p_flip=0, color_jitter_params=None, long_mask=True))
    img, mask, name = ds[0]
    assert img.shape == (3, 32, 32) and img.dtype == torch.float32 and 0 <= img.min() and img.max() <= 1
    assert mask.shape == (32, 32) and mask.dtype == torch.int64 and set(mask.unique().tolist()) <= {0, 1}
    raw = imread(os.path.join(root, "img", name))
    assert np.allclose(img.numpy(), raw.transpose(2, 0, 1) / 255.0)           # BGR order, [0,1], no normalisation
    lab = imread(os.path.join(root, "labelcol", name), gray=True)
    assert np.array_equal(mask.numpy(), (lab > 127).astype(np.int64))
    flip = utils.JointTransform2D(crop=(16, 16), p_flip=1.0, color_jitter_params=None, long_mask=True)
    fi, fm = flip(raw, lab[:, :, None])
    assert fi.shape == (3, 16, 16) and fm.shape == (16, 16)
    g = utils_gray.ImageToImage2D(make_synthetic_dataset(str(tmp_path / "g"), 2, 32, 2, gray=True),
                                  utils.JointTransform2D(crop=None, p_flip=0, color_jitter_params=None, long_mask=True))
    gi, gm, _ = g[0]
    assert gi.shape == (1, 32, 32) and gm.shape == (32, 32)


@pytest.mark.gpu
def test_train_then_test_cli_roundtrip(tmp_path, device):
    """BASELINE.json config 1 plumbing on the GPU: 16 synthetic images, 2 epochs of gatedaxialunet, checkpoint, test.py."""
    env = dict(os.environ, PYTHONPATH=PKG)
    d = str(tmp_path / "data")
    out = str(tmp_path / "run")
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "--train_dataset", d, "--val_dataset", d,
                        "--direc", out, "--batch_size", "4", "--epoch", "2", "--save_freq", "1", "--modelname",
                        "gatedaxialunet", "--learning_rate", "0.001", "--imgsize", "128", "--gray", "no",
                        "--synthetic", "16"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Total_params: 1326850" in r.stdout and "epoch [1/2], loss:" in r.stdout
    ckpt = os.path.join(out, "1", "gatedaxialunet.pth")
    assert os.path.exists(ckpt) and os.path.exists(out + "final_model.pth")
    assert len(os.listdir(os.path.join(out, "1"))) == 16 + 1
    res = str(tmp_path / "res")
    r = subprocess.run([sys.executable, os.path.join(PKG, "test.py"), "--loaddirec", ckpt, "--val_dataset", d, "--direc",
                        res, "--batch_size", "1", "--modelname", "gatedaxialunet", "--imgsize", "128", "--gray", "no"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(os.listdir(res)) == 16
    # --gather: 16 images in replays of 3 (a padded last batch) and one image per replay (the reference's loop) write the same maps
    import numpy as np
    from PIL import Image
    outs = {}
    for gth in ("3", "1"):
        rg = str(tmp_path / ("res" + gth))
        r = subprocess.run([sys.executable, os.path.join(PKG, "test.py"), "--loaddirec", ckpt, "--val_dataset", d, "--direc",
                            rg, "--batch_size", "1", "--modelname", "gatedaxialunet", "--imgsize", "128", "--gray", "no",
                            "--gather", gth], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[gth] = {f: np.asarray(Image.open(os.path.join(rg, f))) for f in sorted(os.listdir(rg))}
        assert len(outs[gth]) == 16
    diff = sum(int((outs["3"][f] != outs["1"][f]).sum()) for f in outs["1"])
    assert diff <= 4, diff                     # (pixels whose logit sits within rounding of the 0.5 threshold)


def test_prefetcher_cpu_passthrough(tmp_path):
    """On a CPU device the prefetcher is plain iteration: same batches, same order (np.random-driven flips included)."""
    from torch.utils.data import DataLoader
    from medt_amd.data import DevicePrefetcher, make_synthetic_dataset
    import utils
    root = make_synthetic_dataset(str(tmp_path / "d"), n=6, size=16, seed=5)
    tf = utils.JointTransform2D(crop=None, p_flip=0.5, color_jitter_params=None, long_mask=True)
    ds = utils.ImageToImage2D(root, tf)
    np.random.seed(3000)
    want = [(x.clone(), y.clone(), n) for x, y, n in DataLoader(ds, batch_size=2, shuffle=False)]
    np.random.seed(3000)
    got = list(DevicePrefetcher(DataLoader(ds, batch_size=2, shuffle=False), "cpu"))
    assert len(got) == len(want) == 3
    for (x0, y0, n0), (x1, y1, n1) in zip(want, got):
        assert torch.equal(x0, x1) and torch.equal(y0, y1) and list(n0) == list(n1)


@pytest.mark.gpu
def test_prefetcher_gpu_matches_blocking_copies(tmp_path, device):
    """Pinned staging + copy stream deliver exactly what `.to(device)` would, in order, with reused staging buffers."""
    from torch.utils.data import DataLoader
    from medt_amd.data import DevicePrefetcher, make_synthetic_dataset
    import utils
    root = make_synthetic_dataset(str(tmp_path / "d"), n=14, size=32, seed=6)
    tf = utils.JointTransform2D(crop=None, p_flip=0.5, color_jitter_params=None, long_mask=True)
    ds = utils.ImageToImage2D(root, tf)
    np.random.seed(3000)
    want = [(x.clone(), y.clone()) for x, y, _ in DataLoader(ds, batch_size=2, shuffle=False)]
    np.random.seed(3000)
    n = 0
    for (x0, y0), (x1, y1, _) in zip(want, DevicePrefetcher(DataLoader(ds, batch_size=2, shuffle=False), device, depth=2)):
        assert x1.is_cuda and y1.is_cuda
        assert torch.equal(x0, x1.cpu()) and torch.equal(y0, y1.cpu())
        n += 1
    assert n == 7
