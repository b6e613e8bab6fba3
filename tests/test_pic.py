"""ParticleInCell semantics of the oracle: index/weights (ParticleInCell.jl:58-71), drop / wrap
(:341-376,444-466), construct_loop order, particle<->node transforms (core_2D.jl:69-128), and
conservation in the propagation-only scenario of tests/T03_PIC_propagation_2d_blob.jl."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from picles_amd import _capi as K, configs
from helpers import make_model


def _iw(zp, i):
    idx = (C.c_int64 * 2)()
    w = np.zeros(2)
    O.lib("libm").picles_oracle_index_weight(zp, i, idx, K.dptr(w))
    return list(idx), w


def test_index_weight_rounding_and_sum():
    for zp in (0.0, 0.25, -0.25, 0.9999996, -1e-9, 1.5, -2.75, 0.1234565, 0.1234575):
        idx, w = _iw(zp, 10)
        b = int(np.floor(zp))
        assert idx == [10 + b, 11 + b]
        assert w[1] == round(zp - np.floor(zp), 6) or abs(w[1] - round(zp - np.floor(zp), 6)) < 1e-15
        assert w[0] == 1.0 - w[1]
    # round half to even at the 7th digit: rint((zp-b)*1e6)/1e6
    idx, w = _iw(0.0000005, 0)
    assert w[1] in (0.0, 1e-6)


def test_particle_node_roundtrip():
    rng = np.random.default_rng(3)
    L = O.lib("libm")
    for _ in range(200):
        z = np.array([rng.uniform(-12, 1), rng.uniform(-3, 3), rng.uniform(-3, 3), 0.0, 0.0])
        c = np.zeros(3)
        z2 = np.zeros(5)
        L.picles_oracle_particle_to_charge(K.dptr(z), K.dptr(c))
        L.picles_oracle_charge_to_particle(K.dptr(c), K.dptr(z2))
        assert np.allclose(z2[:3], z[:3], rtol=1e-13, atol=1e-14)
        assert c[0] == pytest.approx(np.exp(z[0]), rel=4e-16)


def _blob_model(periodic_grid, backend=("libm", 0)):
    cfg = configs.bench06_box(n=16, dx=1000.0, periodic_grid=periodic_grid)
    s = cfg.model["ODEsys"]
    s.input = s.dissipation = s.peak_shift = s.direction = False   # propagation only
    return make_model(cfg, backend), cfg


@pytest.mark.parametrize("periodic", [True, False])
def test_scatter_conserves_and_places(periodic):
    m, cfg = _blob_model(periodic)
    b = m.backend
    b.set_winds(np.zeros((16, 16)), np.zeros((16, 16)))
    z = np.zeros((16, 16, 5))
    on = np.zeros((16, 16), dtype=np.uint8)
    z[..., 0], z[..., 1], z[..., 2] = -3.0, 1.0, 1.0   # harmless values for off particles
    # a blob of particles moving +x/+y by 0.3 / 0.45 cells per step
    for (i, j) in [(5, 5), (6, 5), (5, 6), (14, 14), (15, 15)]:
        on[i, j] = 1
        z[i, j, :3] = [-2.0 + 0.1 * i, 0.5, 0.75]
    b.set_particles(z, on)
    b.zero_state()
    b.advance(600.0)
    S = b.get_state()
    e_in = np.exp(z[..., 0])[on == 1]
    if periodic:
        assert S[..., 0].sum() == pytest.approx(e_in.sum(), rel=1e-14)
    else:
        # (15,15) loses its +x / +y corners over the edge (drop rule)
        assert S[..., 0].sum() < e_in.sum()
        inner = np.exp(-2.0 + 0.1 * 5) * 2 + np.exp(-2.0 + 0.1 * 6)
        assert S[4:9, 4:9, 0].sum() == pytest.approx(inner, rel=1e-14)
    # particle (5,5): x = 0.5*600/1000 = 0.3, y = 0.45 -> corners (5,5),(6,5),(5,6),(6,6)
    zz, _, _, _ = b.get_particles()
    assert zz[5, 5, 3] == pytest.approx(0.3, abs=1e-12) and zz[5, 5, 4] == pytest.approx(0.45, abs=1e-12)
    if periodic:
        # (15,15) wraps onto node 0 in both axes
        assert S[0, 0, 0] > 0


def test_negative_displacement_and_wrap_index():
    m, cfg = _blob_model(True)
    b = m.backend
    b.set_winds(np.zeros((16, 16)), np.zeros((16, 16)))
    z = np.zeros((16, 16, 5)); z[..., 0] = -3.0; z[..., 1] = 1.0
    on = np.zeros((16, 16), dtype=np.uint8)
    on[0, 0] = 1
    z[0, 0, :3] = [0.0, -2.5, -0.5]        # x = -1.5 cells, y = -0.3 cells
    b.set_particles(z, on)
    b.zero_state(); b.advance(600.0)
    S = b.get_state()[..., 0]
    # floor(-1.5) = -2 -> nodes -2,-1 -> wrapped 14,15 ; floor(-0.3) = -1 -> nodes -1,0 -> 15,0
    nz = {(i, j) for i, j in zip(*np.nonzero(S))}
    assert nz == {(14, 15), (15, 15), (14, 0), (15, 0)}
    assert S[14, 15] == pytest.approx(0.5 * 0.3) and S[15, 0] == pytest.approx(0.5 * 0.7)
    assert b.get_counters()["max_reach"] == 2
