"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol
include/mgym.h declares, and fails loudly (no CPU fallback) when no GPU is visible."""
import ctypes as C
import os
import re

import pytest

import modurl_gym_amd as mg
from modurl_gym_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgym.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgym_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"libmgym.so does not export {name}"
    # and the ctypes prototype table covers the header exactly
    assert sorted(_lib.PROTOTYPES) == names


def test_abi_version_and_defaults():
    lib = _lib.load()
    assert lib.mgym_abi_version() == 4
    cfg = _lib.Config()
    assert lib.mgym_default_config(mg.LUNARLANDER, C.byref(cfg)) == _lib.OK
    # reference builder defaults: cartpole.rs:39-40, mountain_car.rs:33, lunar_lander.rs:282-285
    assert (cfg.sutton_barto_reward, cfg.is_euler, cfg.goal_velocity) == (0, 1, 0.0)
    assert (cfg.gravity, cfg.enable_wind, cfg.wind_power, cfg.turbulence_power) == (-10.0, 0, 15.0, 1.5)
    assert cfg.struct_size == C.sizeof(_lib.Config)
    assert lib.mgym_default_config(99, C.byref(cfg)) == _lib.ERR_BAD_ARG


def test_space_metadata_matches_reference():
    s = mg.get_spec(mg.CARTPOLE)  # cartpole.rs:58-69
    assert (s.obs_dim, s.n_actions) == (4, 2)
    assert abs(s.obs_high[0] - 4.8) < 1e-6 and abs(s.obs_high[2] - 0.41887903) < 1e-7 and s.obs_high[1] == float("inf")
    s = mg.get_spec(mg.MOUNTAINCAR)  # mountain_car.rs:42-48
    assert (s.obs_dim, s.n_actions) == (2, 3)
    assert [round(v, 4) for v in (s.obs_low[0], s.obs_low[1], s.obs_high[0], s.obs_high[1])] == [-1.2, -0.07, 0.6, 0.07]
    s = mg.get_spec(mg.LUNARLANDER)  # lunar_lander.rs:1169-1200
    assert (s.obs_dim, s.n_actions) == (8, 4)
    assert list(s.obs_low[6:8]) == [0.0, 0.0] and list(s.obs_high[:4]) == [2.5, 2.5, 10.0, 10.0]
    s = mg.get_spec(mg.MOUNTAINCAR_CONT)
    assert s.action_is_float == 1 and (s.action_low, s.action_high) == (-1.0, 1.0)


def test_bad_config_is_rejected_before_touching_the_gpu():
    # lunar_lander.rs:292-296: gravity must be in (-12, 0)
    with pytest.raises(mg.BadConfigError):
        mg.VecEnv(mg.LUNARLANDER, 4, gravity=-12.0)
    with pytest.raises(mg.BadConfigError):
        mg.VecEnv(mg.LUNARLANDER, 4, gravity=0.0)


@pytest.mark.skipif(mg.device_count() > 0, reason="a GPU is visible")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(mg.MgymError) as ei:
        mg.VecEnv(mg.CARTPOLE, 8)
    assert ei.value.status == _lib.ERR_NO_DEVICE


def test_rust_ffi_declarations_cover_the_header():
    """rust/modurl_gym_mgym/src/sys.rs (source-only: no rustc in the image) must declare every header symbol,
    with the struct fields of mgym_config / mgym_spec in header order."""
    sys_rs = open(os.path.join(ROOT, "rust", "modurl_gym_mgym", "src", "sys.rs")).read()
    assert sorted(set(re.findall(r"pub fn (mgym_[a-z0-9_]+)\s*\(", sys_rs))) == declared_symbols()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mgym.h")).read(), flags=re.S)
    for struct in ("mgym_config", "mgym_spec"):
        c_body = re.search(r"typedef struct %s \{(.*?)\}" % struct, header, re.S).group(1)
        c_fields = [re.sub(r"\[\d+\]", "", f.strip()) for decl in re.findall(r"[a-z0-9_]+_t\s+([^;]+);|float\s+([^;]+);", c_body)
                    for part in decl if part for f in part.split(",")]
        r_body = re.search(r"pub struct %s \{(.*?)\}" % struct, sys_rs, re.S).group(1)
        assert re.findall(r"pub ([a-z0-9_]+):", r_body) == c_fields, struct
    for name, val in re.findall(r"(MGYM_[A-Z_]+) = (\d+)", header):
        m = re.search(r"pub const %s: [a-z_0-9]+ = (\d+);" % name, sys_rs)
        assert m and m.group(1) == val, name


def test_rust_shim_has_the_reference_builder_surface():
    """The reference is constructed as `CartPoleV1::builder().sutton_barto_reward(..).is_euler(..).build()` /
    `::default()` (cartpole.rs:34-44,228-231), `MountainCarV0::builder()` (mountain_car.rs:25-34) and
    `LunarLanderV3::builder()` with `seed: Option<u64>` (lunar_lander.rs:278-291).  The shim cannot be compiled here
    (no rustc), so check its source for the same `#[bon] #[builder]` constructors, argument names, defaults, the gravity
    assert message and `impl Default`."""
    src = open(os.path.join(ROOT, "rust", "modurl_gym_mgym", "src", "lib.rs")).read()
    cargo = open(os.path.join(ROOT, "rust", "modurl_gym_mgym", "Cargo.toml")).read()
    assert re.search(r'^bon = "3', cargo, re.M)
    expect = {
        "CartPoleV1": [("device", "&Device", "&Device::Cpu"), ("sutton_barto_reward", "bool", "false"), ("is_euler", "bool", "true")],
        "MountainCarV0": [("device", "&Device", "&Device::Cpu"), ("goal_velocity", "f32", "0.0")],
        "LunarLanderV3": [("gravity", "f32", "-10.0"), ("enable_wind", "bool", "false"), ("wind_power", "f32", "15.0"),
                          ("turbulence_power", "f32", "1.5"), ("device", "Device", "Device::Cpu")],
    }
    for name, args in expect.items():
        m = re.search(r"#\[bon\]\s*impl %s \{(.*?)\n\}" % name, src, re.S)
        assert m, name
        body = m.group(1)
        assert "#[builder]" in body and "pub fn new(" in body
        for arg, ty, default in args:
            assert re.search(r"#\[builder\(default = %s\)\]\s*%s: %s," % (re.escape(default), arg, re.escape(ty)), body), (name, arg)
        assert re.search(r"impl Default for %s \{\s*fn default\(\) -> Self \{\s*%s::builder\(\)\.build\(\)" % (name, name), src), name
        assert re.search(r"seed: Option<u64>,", body), name
    assert '"gravity (current value: {}) must be between -12 and 0"' in src          # lunar_lander.rs:292-296
    assert "-12.0 < gravity && gravity < 0.0" in src
    # the reference's own unit tests construct the envs like this (cartpole.rs:366, lunar_lander.rs:1596,1693)
    for call in ("CartPoleV1::builder().build()", "CartPoleV1::default()", "MountainCarV0::builder().build()",
                 "LunarLanderV3::builder().enable_wind(true).seed(42).build()"):
        assert call in src, call
