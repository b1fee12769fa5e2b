"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol
include/sendslam_orb.h declares, refuses to run without a HIP device (no CPU fallback), and
the product's constants equal the oracle's."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import pyref
from send_slam_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sendslam_orb.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "send-slam_amd"), "-s"])
    lib = binding.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sendslam_orb.h but not exported"
    assert sorted(binding.EXPORTS) == names
    assert lib.ss_abi_version() == 5


def test_struct_layouts_match_header():
    assert C.sizeof(binding.OrbParams) == 36
    assert C.sizeof(binding.Camera) == 16 + 8 * 8 + 8 + 8 + 8 + 24
    assert binding.KP_DTYPE.itemsize == 24
    assert C.sizeof(binding.StageStats) == 32 + 8 + 24 + 8
    assert C.sizeof(binding.Pose) == 8 + 8 + 24 + 32 + 16
    assert C.sizeof(binding.BatchView) == 8 + 5 * 8
    assert C.sizeof(binding.PipeConfig) == 40 and C.sizeof(binding.PipeSlot) == 32
    assert C.sizeof(binding.PipeResult) == 16 + 8 + 11 * 8
    p = binding.default_params()
    # reference YAML literals, orbslam3_mono_networked.cc:193-206
    assert (p.n_features, p.n_levels, p.ini_th_fast, p.min_th_fast, p.max_batch, p.steer_fma) == (1250, 8, 20, 7, 1, 0)
    assert abs(p.scale_factor - 1.2) < 1e-6 and (p.lapping_x0, p.lapping_x1) == (0, 1000)


def test_no_device_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.OrbError) as e:
        binding.OrbContext(0)
    assert e.value.code == binding.SS_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_product_constants_equal_oracle_constants():
    prod = open(os.path.join(ROOT, "send-slam_amd/csrc/ss_constants.h")).read()
    vals = [int(v) for v in re.findall(r"-?\d+", prod[prod.index("SS_BIT_PATTERN_31_VALUES") + 24:].replace("\\\n", " ").split("#endif")[0])]
    assert vals == pyref.parse_c_int_table(os.path.join(ROOT, "oracle/orb_constants.h"), "ORC_BIT_PATTERN_31")
    orc = open(os.path.join(ROOT, "oracle/orb_constants.h")).read()

    def macro(text, name):
        m = re.search(r"#define\s+" + name + r"\s+(.+)", text)
        return m.group(1).split("/*")[0].strip()

    pairs = [("PATCH_SIZE", "PATCH_SIZE"), ("EDGE_THRESHOLD", "EDGE_THRESHOLD"), ("CELL_W", "CELL_W"),
             ("DEFAULT_NFEATURES", "DEFAULT_NFEATURES"), ("DEFAULT_NLEVELS", "DEFAULT_NLEVELS"),
             ("DEFAULT_INI_TH", "DEFAULT_INI_TH"), ("DEFAULT_MIN_TH", "DEFAULT_MIN_TH"),
             ("DEFAULT_LAPPING_X1", "DEFAULT_LAPPING_X1"), ("GRAY_RY", "GRAY_RY"), ("GRAY_GY", "GRAY_GY"),
             ("GRAY_BY", "GRAY_BY"), ("GRAY_SHIFT", "GRAY_SHIFT"), ("TH_LOW", "TH_LOW"), ("TH_HIGH", "TH_HIGH"),
             ("RESIZE_COEF_BITS", "RESIZE_COEF_BITS")]
    for a, b in pairs:
        assert macro(prod, "SS_" + a) == macro(orc, "ORC_" + b), a
    taps = [int(macro(prod, f"SS_GAUSS_K{i}")) for i in range(4)]
    assert taps + taps[2::-1] == [int(v) for v in re.findall(r"\d+", macro(orc, "ORC_GAUSS_TAPS"))]


def test_nif_glue_type_checks_against_the_c_abi():
    """send-slam_amd/nif/sendslam_nif.c cannot be built here (no OTP); it is at least type-checked against
    include/sendslam_orb.h with a declarations-only erl_nif.h, and must bind the entry points INTEGRATION.md lists."""
    nif = os.path.join(ROOT, "send-slam_amd/nif/sendslam_nif.c")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "tests/native/erl_nif_decls"), "-I" + os.path.join(ROOT, "include"), nif])
    text = open(nif).read()
    for call in ["ss_create(", "ss_destroy(", "ss_set_calibration(", "ss_extract(", "ss_match(", "ss_track("]:
        assert call in text, call
    ex = open(os.path.join(ROOT, "send-slam_amd/nif/hip_backend.ex")).read()
    for name, arity in re.findall(r'\{"(\w+)", (\d+), nif_\w+, ERL_NIF_DIRTY_JOB_CPU_BOUND\}', text):
        m = re.search(r"def " + name + r"\(([^)]*)\), do: :erlang.nif_error", ex)
        assert m and len(m.group(1).split(",")) == int(arity), f"{name}/{arity} has no matching Elixir stub"


def test_geometry_tables_hold_what_the_kernels_assume(tmp_path):
    """Sweep of ss_build_geometry over ~30k image sizes and three scale factors (host build of the product
    source): a size either builds or is 'too small'; a 64x32 tile meets at most 3 x 2 cell windows; every
    evaluated pixel maps to exactly one (tile, sub-list) unit of its cell (tests/native/geometry_sweep.cpp)."""
    exe = str(tmp_path / "geometry_sweep")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests/native/geometry_sweep.cpp"),
                           os.path.join(ROOT, "send-slam_amd/csrc/ss_geometry.cpp")])
    out = subprocess.run([exe, "7"], capture_output=True, text=True)
    assert out.returncode == 0 and "built=" in out.stdout, out.stdout[-2000:]
