"""CPU: the HDF5 files of the reference's data path (synthetic.py:258-261, 285-289, 307-348; scripts/inference.py:168-169) through
rho_diffusion_amd/h5io.py - the system libhdf5 bound with ctypes (h5py is not installed).  The file format is what libhdf5 writes,
so these files open in h5py; here the round trip, row reads, dtypes, attributes and the error behaviour are checked, plus the
dataset class's replay bookkeeping with a CPU device."""
import os

import numpy as np
import pytest
import torch

from rho_diffusion_amd import h5io

pytestmark = pytest.mark.skipif(not h5io.available(), reason="libhdf5 not found on this machine")


def test_round_trip_rows_dtypes_attributes(tmp_path):
    p = tmp_path / "cache.h5"
    d = np.random.default_rng(0).random((7, 4, 3, 2)).astype(np.float32)
    l, m = np.arange(7), -np.arange(7, dtype=np.int32)
    h5io.write(p, {"density": d, "l": l, "m": m, "x64": d.astype(np.float64)[:2]}, attrs={"seed": 1616, "scale": 0.5}, mode="x")
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"                        # the HDF5 signature
    assert sorted(h5io.datasets(p)) == ["density", "l", "m", "x64"]
    assert h5io.shape(p, "density") == (7, 4, 3, 2)
    assert np.array_equal(h5io.read(p, "density"), d)
    assert np.array_equal(h5io.read(p, "density", 3), d[3]) and h5io.read(p, "density", 3).shape == (4, 3, 2)
    assert np.array_equal(h5io.read(p, "density", slice(2, 6)), d[2:6])
    assert h5io.read(p, "density", slice(5, 5)).shape == (0, 4, 3, 2)
    assert h5io.read(p, "l").dtype == np.int64 and h5io.read(p, "m").dtype == np.int32 and h5io.read(p, "x64").dtype == np.float64
    assert int(h5io.read(p, "m", -1)) == -6
    assert h5io.read_attr(p, "seed") == 1616 and h5io.read_attr(p, "scale") == 0.5


def test_error_behaviour_follows_h5py(tmp_path):
    p = tmp_path / "a.h5"
    h5io.write(p, {"data": np.zeros((2, 2), np.float32)})
    with pytest.raises(FileExistsError):
        h5io.write(p, {"data": np.zeros(1, np.float32)}, mode="x")                 # h5py.File(mode="x")
    h5io.write(p, {"data": np.ones((3,), np.float32)}, mode="w")                   # mode "w" truncates
    assert h5io.shape(p, "data") == (3,)
    with pytest.raises(KeyError):
        h5io.read(p, "density")
    with pytest.raises(KeyError):
        h5io.read_attr(p, "seed")
    with pytest.raises(IndexError):
        h5io.read(p, "data", 3)
    with pytest.raises(FileNotFoundError):
        h5io.read(tmp_path / "missing.h5", "data")
    with pytest.raises(h5io.H5Error):
        h5io.write(tmp_path / "c.h5", {"z": np.zeros(2, np.complex64)})


def test_sample_writer_of_the_inference_script(tmp_path):
    """scripts/inference.py:168-169: h5f["data"] = pred_images.cpu().numpy()."""
    from rho_diffusion_amd.utils import save_samples_h5
    x = torch.randn(2, 1, 4, 4, 4)
    save_samples_h5(tmp_path / "gen.h5", x)
    assert np.array_equal(h5io.read(tmp_path / "gen.h5", "data"), x.numpy())


def test_dataset_replay_from_hdf5(tmp_path):
    """SphericalHarmonicDataset(h5_path=...) (synthetic.py:258-261, 285-304): length and items come from the file; a missing file is
    the reference's AssertionError; files with or without the channel axis replay to [1, G, G, G] items."""
    from rho_diffusion_amd.data import SphericalHarmonicDataset
    from rho_diffusion_amd.utils import calculate_sha512_embedding
    G, N = 6, 5
    d = np.random.default_rng(1).random((N, G, G, G)).astype(np.float32)
    l, m = np.array([0, 1, 2, 3, 3]), np.array([0, -1, 2, 0, -3])
    h5io.write(tmp_path / "sh.h5", {"density": d, "l": l, "m": m}, attrs={"seed": 7})
    ds = SphericalHarmonicDataset.from_hdf5(str(tmp_path / "sh.h5"), device="cpu")
    assert len(ds) == N and ds.max_l is None
    x, emb = ds[3]
    assert x.shape == (1, G, G, G) and torch.equal(x[0], torch.from_numpy(d[3]))
    assert torch.equal(emb, calculate_sha512_embedding({"l": 3, "m": 0}, l=256))
    data, labels = ds.batch(4)
    assert data.shape == (4, 1, G, G, G) and torch.equal(data[:, 0], torch.from_numpy(d[:4])) and labels.shape == (4, 256)
    data2, _ = ds.batch(4)                                                          # wraps: fewer than 4 rows are left
    assert torch.equal(data2[:, 0], torch.from_numpy(d[:4]))
    h5io.write(tmp_path / "sh5.h5", {"density": d[:, None], "l": l, "m": m})         # with the channel axis already in the file
    assert SphericalHarmonicDataset(None, h5_path=tmp_path / "sh5.h5", device="cpu")[1][0].shape == (1, G, G, G)
    with pytest.raises(AssertionError):
        SphericalHarmonicDataset(3, h5_path=str(tmp_path / "nope.h5"))
    # a replaying dataset re-serialises (to_hdf5) without touching the GPU
    ds.to_hdf5(tmp_path / "copy")
    assert os.path.exists(tmp_path / "copy.h5") and np.array_equal(h5io.read(tmp_path / "copy.h5", "density"), d)
    assert h5io.read_attr(tmp_path / "copy.h5", "seed") == ds.random_seed
    with pytest.raises(FileExistsError):
        ds.to_hdf5(tmp_path / "copy.h5")
