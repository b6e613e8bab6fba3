"""The HDF5 StateStore (reference: src/Simulations/storing.jl:36-62, 109-131, 142-180) on the CPU: file structure as HDF5.jl
writes it for the reference — checked with the HDF5 project's own `h5dump` where the image has one — round trip through libhdf5,
`run(sim, store=True)` with the oracle backend, reset and replace semantics."""
import shutil
import subprocess

import numpy as np
import pytest

from picles_amd import configs, storing
from picles_amd.simulations import Simulation, close_store, init_state_store, push_state_to_storage, reset_simulation, run
from helpers import make_model

try:
    storing.hdf5()
except OSError as e:          # no libhdf5 in this image
    pytest.skip(str(e), allow_module_level=True)

H5DUMP = shutil.which("h5dump") or shutil.which("h5dump", path="/opt/conda/bin")


def test_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(3)
    t, x, y = np.arange(5) * 600.0, np.linspace(0, 1e5, 7), np.linspace(0, 5e4, 4)
    st = storing.StateStore(tmp_path, t, x, y)
    planes = [np.asfortranarray(rng.random((7, 4, 3))) for _ in t]
    for p in planes[:3]:
        st.write(p)
    st.write(planes[4], i=4)                       # push_state_to_storage!(sim; i)
    st.write(np.ascontiguousarray(planes[3]))       # a C-ordered array lands in the same place
    assert st.iteration == 4
    with pytest.raises(IndexError):
        st.write(planes[0], i=5)
    with pytest.raises(ValueError):
        st.write(np.zeros((4, 7, 3)))
    st.add_winds_forcing({"u": rng.random((5, 7, 4)), "v": None}, {"time": t, "x": x, "y": y})
    st.close()
    st.close()                                      # idempotent
    with pytest.raises(ValueError):
        st.write(planes[0])
    r = storing.read_state_store(tmp_path / "state.h5")
    assert r["data"].shape == (5, 7, 4, 3)
    np.testing.assert_array_equal(r["data"], np.stack(planes))
    assert r["dims"] == ["time", "x", "y", "state"] and r["var_names"] == ["e", "m_x", "m_y"] and r["state"] == ["e", "m_x", "m_y"]
    np.testing.assert_array_equal(r["time"], t); np.testing.assert_array_equal(r["x"], x); np.testing.assert_array_equal(r["y"], y)


@pytest.mark.skipif(H5DUMP is None, reason="no h5dump in this image")
def test_file_structure_seen_by_h5dump(tmp_path):
    """what a reader of the reference's files expects: group `waves`, float64 `data` of (state, y, x, time) in the file's row-major
    terms (= Julia's [time, x, y, state]), vlen UTF-8 strings for `dims` / `var_names`, and the values where they belong"""
    t, x, y = np.arange(2) * 600.0, np.arange(3) * 1e3, np.arange(2) * 2e3
    st = storing.StateStore(tmp_path, t, x, y)
    a = np.zeros((3, 2, 3)); a[2, 1, 0] = 42.5; a[0, 0, 2] = -7.25
    st.write(np.zeros((3, 2, 3))); st.write(a)
    st.close()
    hdr = subprocess.run([H5DUMP, "-H", str(tmp_path / "state.h5")], check=True, capture_output=True, text=True).stdout
    assert 'GROUP "waves"' in hdr and 'DATASET "data"' in hdr and "H5T_IEEE_F64LE" in hdr
    assert "DATASPACE  SIMPLE { ( 3, 2, 3, 2 ) / ( 3, 2, 3, 2 ) }" in hdr
    assert "STRSIZE H5T_VARIABLE" in hdr and "CSET H5T_CSET_UTF8" in hdr
    for name in ("time", "x", "y", "state", "var_names"):
        assert f'DATASET "{name}"' in hdr
    out = subprocess.run([H5DUMP, "-a", "/waves/dims", "-d", "/waves/var_names", str(tmp_path / "state.h5")], check=True,
                         capture_output=True, text=True).stdout
    assert '"time", "x", "y", "state"' in out and '"e", "m_x", "m_y"' in out
    # data[time=1, x=2, y=1, state=0] = 42.5 -> file index (0, 1, 2, 1); data[1, 0, 0, 2] = -7.25 -> (2, 0, 0, 1)
    one = subprocess.run([H5DUMP, "-d", "/waves/data", "-s", "0,1,2,1", "-c", "1,1,1,1", str(tmp_path / "state.h5")], check=True,
                         capture_output=True, text=True).stdout
    assert "42.5" in one
    two = subprocess.run([H5DUMP, "-d", "/waves/data", "-s", "2,0,0,1", "-c", "1,1,1,1", str(tmp_path / "state.h5")], check=True,
                         capture_output=True, text=True).stdout
    assert "-7.25" in two


def test_run_store_equals_cash_store_with_the_oracle_backend(tmp_path):
    """run!(sim; store=true) and run!(sim; cash_store=true) (run.jl:54-112) hold the same states; time axis to stop_time + Δt"""
    cfg = configs.example_00_minimal(n=21, L=40e3)
    a = make_model(cfg, ("pmath", 1))
    sim = Simulation(a, Δt=cfg.Δt, stop_time=cfg.stop_time)
    init_state_store(sim, tmp_path, format="hdf5")
    assert sim.store.format == "hdf5" and sim.store.shape[1:] == (21, 21, 3)
    run(sim, store=True)
    b = make_model(configs.example_00_minimal(n=21, L=40e3), ("pmath", 1))
    sim2 = Simulation(b, Δt=cfg.Δt, stop_time=cfg.stop_time)
    run(sim2, cash_store=True)
    r = storing.read_state_store(tmp_path / "state.h5")
    n = len(sim2.store.store)
    assert n == int(cfg.stop_time // cfg.Δt) + 2           # initial state + one step past stop_time
    for k in range(n):
        np.testing.assert_array_equal(r["data"][k], sim2.store.store[k])
    np.testing.assert_array_equal(r["time"][:n], np.arange(n) * cfg.Δt)
    np.testing.assert_allclose(r["x"], np.linspace(0, 40e3, 21))


def test_reset_and_replace(tmp_path):
    cfg = configs.example_00_minimal(n=11, L=20e3)
    sim = Simulation(make_model(cfg, ("pmath", 1)), Δt=cfg.Δt, stop_time=1200.0)
    init_state_store(sim, tmp_path, name="s", format="hdf5")
    run(sim, store=True)                                       # closes the file
    first = storing.read_state_store(tmp_path / "s.h5")["data"].copy()
    assert first[0].any()
    init_state_store(sim, tmp_path, name="s", format="hdf5")   # replace=true: a fresh file
    sim.store.write(first[1])
    reset_simulation(sim)                                      # reset_state_store!: all zero, counter back to the start
    assert sim.store.iteration == 0 and sim.model.clock.time == 0.0 and not np.asarray(sim.model.State).any()
    run(sim, store=True)
    again = storing.read_state_store(tmp_path / "s.h5")["data"]
    assert not again[0].any()                                  # reset_simulation! clears State after re-seeding (run.jl:172-175)
    np.testing.assert_array_equal(again[1:], first[1:])
    sim2 = Simulation(make_model(cfg, ("pmath", 1)), Δt=cfg.Δt, stop_time=1200.0)
    init_state_store(sim2, tmp_path, name="t", format="npy")
    push_state_to_storage(sim2, i=2)
    close_store(sim2)
    assert np.load(tmp_path / "t.waves.data.npy")[2].shape == (11, 11, 3)


def test_format_selection(tmp_path, monkeypatch):
    st = storing.make_state_store(tmp_path, [0.0], [0.0, 1.0], [0.0], format="npy")
    assert st.format == "npy"
    st.close()
    with pytest.raises(ValueError):
        storing.make_state_store(tmp_path, [0.0], [0.0], [0.0], format="zarr")
    monkeypatch.setattr(storing, "_lib", None)
    monkeypatch.setattr(storing, "_find_hdf5", lambda: (_ for _ in ()).throw(OSError("none here")))
    with pytest.raises(OSError):
        storing.make_state_store(tmp_path, [0.0], [0.0], [0.0], format="hdf5")
    with pytest.warns(UserWarning):
        st = storing.make_state_store(tmp_path, [0.0], [0.0], [0.0], format="auto")
    assert st.format == "npy"
