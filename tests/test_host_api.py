"""Host-side mirror of the reference interface: grid/mask classes, ocean_points, run! step
count, movie_time_step! State semantics — computed here by the CPU oracle injected as backend
(the product backend is the HIP library; see test_gpu_*.py)."""
import numpy as np
import pytest

from picles_amd import configs, grids, models, fetch_relations as FR
from picles_amd.simulations import Simulation, run, initialize_simulation
from picles_amd.timesteppers import movie_time_step, time_step, time_step_advance, time_step_remesh
from helpers import make_model, oracle_factory

ORACLE = ("libm", 0)


def test_make_boundaries_matches_reference_rules():
    # mask_utils.jl:38-55: ring = 3 on non-periodic axes, land boundary = 2 next to ocean
    mask = np.ones((6, 5), dtype=bool)
    mask[2:4, 2] = False
    tot = grids.make_boundaries(mask, grids.N_NonPeriodic(6), grids.N_Periodic(5))
    assert (tot[0, :] == 3).all() and (tot[-1, :] == 3).all()
    assert tot[2, 2] == 2 and tot[3, 2] == 2          # land cells touching ocean become 'land boundary'
    assert tot[1, 1] == 1 and tot[2, 0] == 1          # periodic axis: no ring in y
    g = grids.TwoDCartesianGridMesh(100e3, 51, 100e3, 51)
    assert g.stats.dx == 2000.0 and int(g.stats.Nx) == 51
    assert (g.data.mask[1:-1, 1:-1] == 1).all() and (g.data.mask[0] == 3).all()
    assert g.data.x[3, 7] == 6000.0 and g.data.y[3, 7] == 14000.0


def test_ocean_points_order_and_count():
    cfg = configs.example_00_minimal()
    m = make_model(cfg, ORACLE)
    assert len(m.ocean_points) == 49 * 49                 # periodic_boundary = false: mask == 1 only
    assert m.ocean_points[0] == (1, 1) and m.ocean_points[1] == (2, 1)   # column-major findall
    assert m.backend.n_stepped == 49 * 49
    cfg = configs.T04_2D_reg_test(periodic=True)
    m = make_model(cfg, ORACLE)
    assert len(m.ocean_points) == 31 * 31                 # ocean list, then the grid-boundary ring
    assert m.ocean_points[29 * 29] == (0, 0)
    assert m.backend.n_stepped == 31 * 31
    _, _, bnd, _ = m.backend.get_particles()
    assert not bnd.any()                                   # periodic model: only mask == 2 is 'boundary'


def test_run_takes_one_step_past_stop_time():
    cfg = configs.example_00_minimal(n=21, L=40e3)
    m = make_model(cfg, ORACLE)
    sim = Simulation(m, Δt=cfg.Δt, stop_time=cfg.stop_time)
    run(sim, cash_store=True)
    assert len(sim.store.store) == 14                      # initial + 13 steps for 2 h at 10 min (run.jl:113)
    assert m.clock.time == 13 * 600.0
    S0, S1 = sim.store.store[0], sim.store.store[1]
    # initial State = seeded (e, m) on every non-land node, ring included (core_2D.jl:434-488)
    ws = FR.get_initial_windsea(10.0, 10.0, 600.0)
    assert S0[0, 0, 0] == pytest.approx(ws["E"], rel=1e-13) and S0[10, 10, 1] == pytest.approx(ws["m_x"], rel=1e-13)
    # ring particles are never stepped: after a zero-first step the ring only holds scatter spill
    assert S1[10, 10, 0] > S0[10, 10, 0]


def test_movie_time_step_semantics():
    cfg = configs.T04_2D_reg_test(n=15, L=56e3)
    m = make_model(cfg, ORACLE)
    sim = Simulation(m, Δt=cfg.Δt, stop_time=3600.0)
    initialize_simulation(sim)
    S_seed = m.State.copy()
    movie_time_step(m, cfg.Δt)
    M1 = m.MovieState.copy()
    assert np.all(m.State == 0.0)                          # State zeroed after remesh (TimeSteppers.jl:245)
    # first movie step scatters ON TOP of the seeded State (it is only zeroed after the step)
    m2 = make_model(configs.T04_2D_reg_test(n=15, L=56e3), ORACLE)
    initialize_simulation(Simulation(m2, Δt=cfg.Δt, stop_time=3600.0))
    m2.backend.zero_state()
    movie_time_step(m2, cfg.Δt)
    assert np.allclose(M1, m2.MovieState + S_seed, rtol=1e-13, atol=1e-300)


def test_split_advance_remesh_equals_time_step():
    a = make_model(configs.example_00_minimal(n=17, L=32e3), ORACLE)
    b = make_model(configs.example_00_minimal(n=17, L=32e3), ORACLE)
    for m in (a, b):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    for _ in range(3):
        a.backend.zero_state()
        time_step(a, 600.0)
        b.backend.zero_state()
        time_step_advance(b, 600.0)
        time_step_remesh(b, 600.0)
        b.backend.tick(600.0); b.clock.time += 600.0
    assert np.array_equal(a.State, b.State)
    za, zb = a.backend.get_particles()[0], b.backend.get_particles()[0]
    assert np.array_equal(za, zb)


def test_calm_region_switches_particles_off_and_on():
    """tests/T04_2D_on_off_particle_tests.jl idea: wind below sqrt(2) seeds 'off' particles;
    they switch on when the wind picks up (mapping_2D.jl:172-185) and remesh branch D turns
    starved particles off."""
    cfg = configs.growing_decaying_winds(n=24, dx=2000.0, n_steps=4)
    m = make_model(cfg, ORACLE)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    _, on0, _, _ = m.backend.get_particles()
    assert not on0[:12].any() and on0[16:].all()           # calm half seeded off
    for _ in range(4):
        time_step(m, cfg.Δt, zero_first=True)
    _, on1, _, st = m.backend.get_particles()
    assert on1[16:-1, 1:-1].all()
    assert m.backend.get_counters()["particles_advanced"] > 0


class _FakeBackend:
    """the call surface LazyState needs, with the two generation counters of picles_amd.driver.HipModel"""

    def __init__(self, shape):
        self.S = np.ones(shape)
        self.gen = self.state_gen = 0
        self.zero_first_seen = []

    def get_state(self):
        return self.S.copy()

    def set_state(self, s):
        self.gen += 1; self.state_gen += 1
        self.S = np.array(s, dtype=float)

    def set_particles(self):            # touches the device, leaves State alone
        self.gen += 1

    def seed(self):                     # init_particles!: writes the seeds' State
        self.gen += 1; self.state_gen += 1
        self.S = np.full(self.S.shape, 3.0)

    def time_step(self, zero_first):
        self.gen += 1; self.state_gen += 1
        self.zero_first_seen.append(bool(zero_first))
        self.S = (np.zeros_like(self.S) if zero_first else self.S) + 0.5


def test_lazy_state_zero_survives_calls_that_leave_state_alone():
    """`State .= 0; set_particles(...); time_step!` must start from a zeroed State (the recorded zero-fill was
    dropped when ANY backend call bumped the generation), while `State .= 0; init_particles!; time_step!` accumulates onto the
    seeds' State as the reference does (run.jl:199-247 writes State; TimeSteppers.jl:109-166 does not zero it)."""
    b = _FakeBackend((4, 3, 3))
    st = models.LazyState(b, (4, 3, 3))
    st.fill(0.0)
    b.set_particles()
    assert st.before_step() is True                  # the zero-fill still stands
    assert np.all(np.asarray(st) == 0.0)             # and a read sees zeros, not the stale device field
    b.time_step(True); st.after_step()
    assert np.all(np.asarray(st) == 0.5)
    st.fill(0.0)
    b.seed()
    assert np.all(np.asarray(st) == 3.0)             # the seeds' State, read from the device
    assert st.before_step() is False                 # the step accumulates onto it
    b.time_step(False); st.after_step()
    assert np.all(np.asarray(st) == 3.5)
    # a backend that only keeps `gen` (any call invalidates) still never reads a stale mirror
    b2 = _FakeBackend((2, 2, 3)); del b2.state_gen
    st2 = models.LazyState(b2, (2, 2, 3))
    _ = np.asarray(st2)
    b2.set_particles()
    b2.S[:] = 7.0
    assert np.all(np.asarray(st2) == 7.0)


def test_hip_model_state_gen_counts_state_writers_only():
    """the driver's two counters, without a device: which methods bump `state_gen`"""
    import inspect
    from picles_amd.driver import HipModel
    writers = {"seed", "time_step", "run_steps", "advance", "zero_state", "scatter_remesh", "step_rows", "end_fused_step",
               "slab_run_steps", "set_state", "scatter_particles"}
    for name, fn in inspect.getmembers(HipModel, inspect.isfunction):
        src = inspect.getsource(fn)
        assert ("self.state_gen += 1" in src) == (name in writers), name
        if name in writers:
            assert "self.gen += 1" in src, name


def test_windsea_object_behind_the_reference_entry_points():
    """fetch_relations: one object (Windsea) behind get_initial_windsea / MinimalWindsea / MinimalParticle / MinimalState
    (FetchRelations.jl:314-415): the Dict keys of the reference read its fields, the particle form is (ln e, c̄x, c̄y, 0, 0),
    calm winds are evaluated at 0.1 m/s along their own direction, the minimal sea takes a zero component as +1."""
    import math
    ws = FR.get_initial_windsea(10.0, -5.0, 1800.0)
    assert set(ws.keys()) == {"E", "lne", "Hs", "cg_bar_x", "cg_bar_y", "cg_bar", "f_peak", "T_bar", "X_tilde", "m_x", "m_y"}
    assert ws.as_dict() == {k: ws[k] for k in ws.keys()}
    assert ws["lne"] == math.log(ws["E"]) and ws["Hs"] == 4 * math.sqrt(ws["E"])
    assert math.hypot(ws["cg_bar_x"], ws["cg_bar_y"]) == pytest.approx(ws["cg_bar"], rel=1e-15)
    assert ws["cg_bar_y"] / ws["cg_bar_x"] == pytest.approx(-0.5, rel=1e-15)              # along the wind
    assert math.hypot(ws["m_x"], ws["m_y"]) == pytest.approx(ws["E"] / (2 * ws["cg_bar"]), rel=1e-15)
    assert FR.get_initial_windsea(10.0, -5.0, 1800.0, particle_state=True) == [ws["lne"], ws["cg_bar_x"], ws["cg_bar_y"], 0.0, 0.0]
    assert FR.get_initial_windsea(10.0, -5.0, -1800.0)["E"] == ws["E"]                     # |time scale|
    with pytest.raises(KeyError):
        ws["no_such_scale"]
    # calm: the relations at 0.1 m/s, the components keep their size (the reference divides by the floored speed)
    calm = FR.get_initial_windsea(0.03, 0.04, 600.0)
    at_floor = FR.get_initial_windsea(0.06, 0.08, 600.0)
    assert calm["E"] == at_floor["E"] and calm["cg_bar"] == at_floor["cg_bar"]
    assert calm["cg_bar_x"] == pytest.approx(0.5 * at_floor["cg_bar_x"], rel=1e-15)
    # the minimal sea: speed U_MIN along the wind, whatever the wind's speed; a zero component counts as +1
    a, b = FR.MinimalWindsea(10.0, 10.0, 600.0), FR.MinimalWindsea(2.0, 2.0, 600.0)
    assert a.as_dict() == b.as_dict()
    assert FR.MinimalWindsea(0.0, 0.0, 600.0).as_dict() == a.as_dict()
    assert FR.MinimalState(10.0, 10.0, 600.0) == [a["E"], a["m_x"] ** 2 + a["m_y"] ** 2]
    assert FR.MinimalParticle(10.0, 10.0, 600.0) == [a["lne"], a["cg_bar_x"], a["cg_bar_y"], 0, 0]
    # SURVEY Appendix D (hand-computed): MinimalState(·,·,600)
    assert FR.MinimalState(2.0, 2.0, 600.0) == pytest.approx([1.253106339976604e-6, 1.2821164e-9], rel=1e-7)


def test_node_classes_on_small_and_degenerate_masks():
    """grids.make_boundaries / interior_boundary / make_boundary_lists (mask_utils.jl:14-82) on shapes the slices must survive:
    one row, one column, all land, all ocean; the coast is circular in both axes whatever the axis types are"""
    for shape in [(1, 1), (1, 5), (5, 1), (2, 2), (3, 4)]:
        for fill in (True, False):
            m = np.full(shape, fill)
            for NX in (grids.N_Periodic(shape[0]), grids.N_NonPeriodic(shape[0])):
                for NY in (grids.N_Periodic(shape[1]), grids.N_NonPeriodic(shape[1])):
                    t = grids.make_boundaries(m, NX, NY)
                    assert t.dtype == np.int8 and t.shape == shape
                    assert not (t == grids.LAND_BOUNDARY).any()              # no coast without both land and ocean
                    inner = t[(slice(1, -1) if isinstance(NX, grids.N_NonPeriodic) else slice(None)),
                              (slice(1, -1) if isinstance(NY, grids.N_NonPeriodic) else slice(None))]
                    assert (inner == (grids.OCEAN if fill else grids.LAND)).all()
                    lists = grids.make_boundary_lists(t)
                    assert len(lists.ocean) + len(lists.land_boundary) + len(lists.grid_boundary) + int((t == grids.LAND).sum()) == t.size
    m = np.ones((5, 4), dtype=bool)
    m[0, 0] = False                                          # one land node in the corner: it is coast, seen through the wrap too
    assert grids.interior_boundary(m).sum() == 1 and grids.interior_boundary(m)[0, 0]
    m = np.zeros((5, 4), dtype=bool)
    m[4, 3] = True                                           # one ocean node in the opposite corner: its four neighbours (two by wrap) are coast
    cb = grids.interior_boundary(m)
    assert cb.sum() == 4 and cb[3, 3] and cb[0, 3] and cb[4, 2] and cb[4, 0]
    lists = grids.make_boundary_lists(grids.make_boundaries(m, grids.N_Periodic(5), grids.N_Periodic(4)))
    assert lists.ocean == [(4, 3)] and lists.land_boundary == [(4, 0), (4, 2), (0, 3), (3, 3)]      # i fastest
