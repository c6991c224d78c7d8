"""CPU tier: the C-ABI shared library builds for gfx950, loads without a GPU, and exports every
symbol include/mkamd_voxel.h declares.  No GPU compute is called here; without a device the library
must refuse loudly (there is no CPU fallback: the host entry point is reached by name only)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mkamd_voxel.h")


def declared_symbols():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("mkamd_voxel.h", "mkamd_distance.h", "mkamd_xtc.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mkamd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from moleculekit_amd import _build
    return _build.build()


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("mkamd_calculate_occupancy", "mkamd_voxelize_lattice_host", "mkamd_voxelize_lattice_dev",
                 "mkamd_occupancy_centers_host", "mkamd_grid_centers_host", "mkamd_ctx_create", "mkamd_last_error"):
        assert must in syms
    assert "torch" not in open(HEADER).read().split("*/", 1)[1].lower().replace("torch.cuda.current_stream", "")


def test_library_exports_every_declared_symbol(lib_path):
    L = ctypes.CDLL(lib_path)
    for s in declared_symbols():
        assert hasattr(L, s), f"libmkamd.so does not export {s}"


def test_python_binding_covers_every_declared_symbol(lib_path):
    from moleculekit_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_library_contains_gfx950_code_object(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob and b"k_voxelize_tiles" in blob


def test_no_silent_cpu_fallback(lib_path):
    """Without a HIP device every compute entry point must fail loudly."""
    from moleculekit_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible; the refusal path is exercised on CPU-only boxes")
    with pytest.raises(RuntimeError, match="no HIP device|no ROCm"):
        _lib.Context(0)
    import numpy as np
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    for method in ("C", "HIP"):
        with pytest.raises(RuntimeError):
            getVoxelDescriptors(None, boxsize=[4, 4, 4], center=[0, 0, 0], usercoords=np.zeros((1, 3), np.float32),
                                userchannels=np.ones((1, 8)), method=method)
    # the host implementation exists but only by name: nothing above fell back to it
    feats, _, _ = getVoxelDescriptors(None, boxsize=[4, 4, 4], center=[0, 0, 0], usercoords=np.zeros((1, 3), np.float32),
                                      userchannels=np.ones((1, 8)), method="CPU")
    assert feats.shape == (64, 8) and feats.max() > 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "moleculekit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", "").lower() or f == "distributed.py", f
