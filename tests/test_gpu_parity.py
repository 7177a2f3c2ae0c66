"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI (libsmcx.so), against the CPU oracle on identical seeded inputs.

Tolerances: tests/tolerances.py states them ONCE (integers equal; single evaluations TOL.SINGLE; free-running
chains by a schedule in the sweep index derived from the measured rounding drift; observables at north_star's
TOL.OBSERVABLE = 1e-6 relative).  Chains are chaotic, so longer chains are compared through
invariants, not value by value.
"""
import os
import subprocess

import numpy as np
import pytest

import tolerances as TOL
from tolerances import rel

pytestmark = pytest.mark.gpu

T = A = 1.1


def make_engine(S, O, R0, nrep, **kw):
    N = len(R0) // 3 if np.ndim(R0) == 1 else np.shape(R0)[1] // 3
    p = S.default_params(N, nrep, **kw)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    return eng, p


def sys_of(O, p):
    return O.make_sys(p.N, M=p.M, L=p.L, Lz=p.Lz, cutoff=p.cutoff, a0=p.a0, b0=p.b0, Ncx=p.Ncx, Ncz=p.Ncz)


def dense_state(O, N=1024, seed=5):
    """a state with neighbours inside the cutoff, particles at, near and beyond the walls"""
    rs = np.random.RandomState(seed)
    R = O.fcc(8, 4).reshape(-1, 3)[:N].copy()
    R += 0.25 * rs.standard_normal(R.shape)
    R[:, 2] *= 0.5
    R[:40, 2] = 120.0 - np.abs(rs.standard_normal(40)) * 1.5      # close to the upper wall
    R[40:80, 2] = -120.0 + np.abs(rs.standard_normal(40)) * 1.5   # close to the lower wall
    R[80, 2] = 120.0 - 1e-3
    R[81, 2] = -120.0 + 1e-3
    R[82, 2] = 121.5   # beyond the wall: clamp branch SMC.c:738-739
    R[83, 2] = -120.0  # exactly on it
    R[:, 0] -= 33.0 * np.rint(R[:, 0] / 33.0)
    R[:, 1] -= 33.0 * np.rint(R[:, 1] / 33.0)
    return R.ravel()


# ------------------------------------------------------------------ K1-K4
@pytest.mark.parametrize("N", [256, 1024])
def test_eval_moves_matches_oracle(S, O, N):
    R = dense_state(O, 1024)[:3 * N] if N == 1024 else O.fcc(4, 4)
    if N == 256:
        R = R + 0.0
        R[2::3] = R[2::3] * 0 + np.linspace(-119.9, 119.9, 256)  # spread to both walls
        R[0::3] *= 0.2; R[1::3] *= 0.2                            # compress: pairs inside the cutoff
    nrep = 48
    rs = np.random.RandomState(N)
    n = rs.randint(0, N, nrep).astype(np.int32)
    if N == 1024:
        n[:6] = [80, 81, 82, 83, 0, 41]
    Rb = np.tile(R, (nrep, 1))
    prop = np.stack([Rb[r, 3 * n[r]:3 * n[r] + 3] + 0.3 * rs.standard_normal(3) for r in range(nrep)])
    prop[:, 0] -= 33.0 * np.rint(prop[:, 0] / 33.0)
    prop[:, 1] -= 33.0 * np.rint(prop[:, 1] / 33.0)
    prop[1] = [1.0, 2.0, 125.0]   # proposal beyond the wall
    p = S.default_params(N, nrep)
    out = S.eval_moves(p, Rb, O.W_FIXTURE, n, prop)
    s = sys_of(O, p)
    nz = 0
    for r in range(nrep):
        Um, Fm, Un, Fn = O.eval_move(s, Rb[r], O.W_FIXTURE, int(n[r]), prop[r])
        ref = np.array([Um, *Fm, Un, *Fn])
        sc = np.abs(ref).max()
        assert np.all(np.abs(out[r] - ref) <= TOL.SINGLE * (np.abs(ref) + sc)), (r, n[r], out[r], ref)
        nz += np.count_nonzero(ref)
    assert nz > 6 * nrep  # the in-cutoff and wall branches really ran


def test_eval_moves_without_walls(S, O):
    R = dense_state(O, 1024)
    p = S.default_params(1024, 4, flags=S.FLAG_E0_RESTART)
    n = np.array([1, 500, 82, 1023], dtype=np.int32)
    prop = np.stack([R[3 * i:3 * i + 3] + 0.1 for i in n])
    out = S.eval_moves(p, np.tile(R, (4, 1)), None, n, prop)
    s = sys_of(O, p)
    for r in range(4):
        Rr = R.copy()
        Um = O.energy_single(s, Rr, int(n[r])); Fm = O.force_single(s, Rr, int(n[r]))
        Rr[3 * n[r]:3 * n[r] + 3] = prop[r]
        Un = O.energy_single(s, Rr, int(n[r])); Fn = O.force_single(s, Rr, int(n[r]))
        ref = np.array([Um, *Fm, Un, *Fn])
        assert np.all(np.abs(out[r] - ref) <= TOL.SINGLE * (np.abs(ref) + np.abs(ref).max()))


# ------------------------------------------------------------------ K5
@pytest.mark.parametrize("Na,Nz", [(4, 4), (8, 4), (8, 16), (16, 4)])
def test_total_energy_matches_oracle(S, O, Na, Nz):
    R0 = O.fcc(Na, Nz)
    eng, p = make_engine(S, O, R0, 3)
    E = eng.total_energy()
    ref = O.total_energy(sys_of(O, p), R0, O.W_FIXTURE)
    assert np.all(rel(E, ref) < TOL.SINGLE)
    eng.close()


def test_total_energy_dense_state(S, O):
    R = dense_state(O)
    eng, p = make_engine(S, O, R, 2)
    ref = O.total_energy(sys_of(O, p), R, O.W_FIXTURE)
    assert np.all(rel(eng.total_energy(), ref) < TOL.SINGLE) and abs(ref) > 1e30  # clamp term dominates
    eng.close()


@pytest.mark.parametrize("case", ["unconfined_gas", "one_plane", "N4000", "N16384", "N2", "last_index_on_top", "per_replica"])
def test_total_energy_of_states_that_stress_the_z_ranking(S, O, case):
    """smcx_total_energy ranks a replica's particles by an 18-bit z of the replica's OWN z range and tests only ranks within the
    cutoff: states whose range is far wider than the box (no walls: nothing confines z, SMC.c:626-646 has no z wrap), zero (one
    plane), N not a power of two, the largest N of the ranked kernel, the last index on top of the range (its key is the highest
    a particle can have, next to the padding's), and replicas whose ranges differ."""
    rs = np.random.RandomState(11)
    walls, per = True, None
    if case == "unconfined_gas":
        N, walls = 1024, False
        R = O.fcc(8, 4).reshape(-1, 3) + 0.05 * rs.standard_normal((N, 3))       # the lattice keeps pairs inside the cutoff ...
        R[300:, 2] = rs.uniform(-950.0, 950.0, N - 300)                          # ... most particles are far outside the box
        R[:, :2] -= 33.0 * np.rint(R[:, :2] / 33.0)                              # (smcx_upload takes |z| <= 4 Lz = 960)
    elif case == "one_plane":
        N = 256
        g = (np.arange(16) + 0.5) * (33.0 / 16) - 16.5
        R = np.c_[np.stack(np.meshgrid(g, g), -1).reshape(-1, 2) + 0.05 * rs.standard_normal((N, 2)), np.full(N, 7.25)]
    elif case == "N4000":
        N = 4000
        R = O.fcc(10, 10).reshape(-1, 3) + 0.05 * rs.standard_normal((N, 3))
    elif case == "N16384":
        N = 16384
        R = O.fcc(16, 16).reshape(-1, 3) + 0.05 * rs.standard_normal((N, 3))
    elif case == "N2":
        N = 2
        R = np.array([[0.0, 0.0, 0.0], [0.7, 0.8, 0.9]])
    elif case == "last_index_on_top":
        N = 16384
        R = O.fcc(16, 16).reshape(-1, 3).copy()
        R[N - 1] = [1.0, 1.0, R[:, 2].max() + 1.05]
        R[N - 2] = [1.6, 1.0, R[N - 1, 2] - 1e-9]
    else:
        N = 1024
        per = np.stack([O.fcc(8, 4).reshape(-1, 3) * [1.0, 1.0, f] + 0.05 * rs.standard_normal((N, 3)) for f in (1.0, 0.25, 0.02, 1.0)])
    nrep = 4 if per is not None else 2
    p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE if walls else S.FLAG_E0_RESTART)
    eng = S.Engine(p)
    states = per if per is not None else np.stack([R] * nrep)
    eng.upload(states.reshape(nrep, -1) if per is not None else R.ravel(), O.W_FIXTURE if walls else None)
    E = eng.total_energy()
    s = sys_of(O, p)
    for r in range(nrep):
        X = np.ascontiguousarray(states[r].ravel())
        ref = O.total_energy(s, X, O.W_FIXTURE) if walls else O.lib().orc_energy(O.C.byref(s), O._ptr(X))
        assert abs(ref) > 1e-3 and rel(E[r], ref) < TOL.SINGLE, (case, r, E[r], ref)
    eng.close()


# ------------------------------------------------------------------ S1: one sweep
@pytest.mark.parametrize("case", ["N256", "N1024", "N108_padded", "N1024_dense_x2waves", "N4096"])
def test_single_sweep_matches_oracle(S, O, case):
    kw = {}
    if case == "N256":
        R0 = O.fcc(4, 4)
    elif case == "N1024":
        R0 = O.fcc(8, 4)
    elif case == "N108_padded":
        R0, _ = O.box_ref(108, 33.0, 200.0); kw = dict(Lz=200.0)
    elif case == "N1024_dense_x2waves":
        R0 = dense_state(O); R0[3 * 82 + 2] = 119.0; kw = dict(tune_slots=16, tune_waves=2)
    else:
        R0 = O.fcc(8, 16)
    nrep = 3
    eng, p = make_engine(S, O, R0, nrep, **kw)
    eng.run(0, 1, 1)
    ob = eng.observables()
    Rg = eng.positions()
    s = sys_of(O, p)
    for r in range(nrep):
        R = np.array(R0, copy=True)
        E0 = O.total_energy(s, R, O.W_FIXTURE)
        acc, E1, _ = O.sweep(s, O.Rng(12345 + r), R, O.W_FIXTURE, A, T, E=E0)
        assert int(ob["accepted"][r]) == acc
        TOL.assert_energy(ob["E_last"][r], E1, 1, case)
        TOL.assert_positions(Rg[r], R, 1, case)
        assert ob["zhist"][r].sum() == p.N  # gather_lapse 1: one histogram of the initial state
    eng.close()


def test_compat_shim_one_particle_moves(S, O):
    """smcx_one_particle_moves keeps the contract of SMC.h:102 (in-place R/Rn, += j, += U)"""
    R0 = O.fcc(4, 4)
    p = S.default_params(256, 1)
    s = sys_of(O, p)
    rng_gpu = S.rng_seed(777)
    rng_cpu = O.Rng(777)
    Rg, Rc = R0.copy(), R0.copy()
    Rn = np.zeros_like(Rg)
    j, U = 5, 1.25
    jc, Uc = 5, 1.25
    for k in range(3):
        j, U = S.one_particle_moves(p, rng_gpu, Rg, Rn, O.W_FIXTURE, 2.2, T, j, U)
        a, Uc, _ = O.sweep(s, rng_cpu, Rc, O.W_FIXTURE, 2.2, T, E=Uc)
        jc += a
        assert j == jc
        TOL.assert_energy(U, Uc, k + 1)
        TOL.assert_positions(Rg, Rc, k + 1)
        assert np.array_equal(Rn, Rg)
    # the explicit RNG handle advanced by exactly 3*(4N+1) draws
    h = [int(v) for v in rng_gpu[:31]]
    left = int(rng_gpu[31])
    nxt_gpu = (h[31 - left] >> 1) if left else (((h[0] + h[28]) & 0xFFFFFFFF) >> 1)
    assert nxt_gpu == rng_cpu.rand()


# ------------------------------------------------------------------ C: chains
@pytest.mark.parametrize("Na,Nz,nsw,kw", [(4, 4, 20, {}), (8, 4, 10, {}),
                                          (8, 4, 10, dict(tune_slots=16, tune_waves=4)),
                                          (4, 4, 20, dict(tune_slots=16, tune_waves=1))])
def test_free_running_chain_observables(S, O, Na, Nz, nsw, kw):
    """Free-running chains inside the chaos horizon: 20 sweeps at N=256 (dilute start),
    10 at N=1024 (denser: a 1e-16 perturbation reaches 1e-5 by sweep ~18 and flips an
    accept decision, measured on the GPU in round 1)."""
    R0 = O.fcc(Na, Nz)
    nrep = 4
    eng, p = make_engine(S, O, R0, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, **kw)
    eng.run(0, nsw, 1)
    ob = eng.observables()
    Es, jj = eng.series(nsw)
    g, oob = eng.hist_info()
    s = sys_of(O, p)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, nsw, 1)
        assert rel(ob["acceptance_ratio"][r], ref["acceptance_ratio"]) < TOL.OBSERVABLE
        assert rel(ob["meanE"][r], ref["meanE"]) < TOL.OBSERVABLE
        assert rel(ob["dE"][r], ref["dE"]) < 10 * TOL.OBSERVABLE       # (a standard deviation: a difference of nearby numbers)
        prof_g = ob["zhist"][r] / float(g[r]); prof_c = ref["zhist"] / float(ref["gathers"])
        assert np.abs(prof_g - prof_c).sum() <= TOL.OBSERVABLE * prof_c.sum() + 2.0 / ref["gathers"]
        assert int(g[r]) == nsw and int(oob[r]) == 0 and ob["zhist"][r].sum() == nsw * p.N
        # pointwise the trajectories separate exponentially (chaos): the schedule of
        # tests/tolerances.py is tight over the first sweeps and at its cap ("no pair dropped") from the fourth;
        # the averaged observables above stay within the north-star 1e-6
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        assert np.array_equal(jj[r][:8], ref["jj"][:8])
    eng.close()


def test_benchmark_kernel_against_oracle_N4096(S, O):
    """the kernel bench.py measures (N=4096: 64 particles per lane, one wavefront per replica, x,y as
    int16 and z as int16 screening copies) against the CPU oracle, 3 sweeps from the benchmark's start"""
    R0 = O.fcc(8, 16)
    nsw, nrep = 3, 2
    eng, p = make_engine(S, O, R0, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=64, tune_waves=1)
    assert eng.kernel_form == (2, "smcx::sweep_kernel_mc64"), eng.kernel_form
    eng.run(0, nsw, 1)
    ob = eng.observables()
    Es, jj = eng.series(nsw)
    s = sys_of(O, p)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, nsw, 1)
        assert np.array_equal(jj[r], ref["jj"])
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        TOL.assert_mean_energy(ob["meanE"][r], ref["meanE"], nsw)
        assert rel(ob["acceptance_ratio"][r], ref["acceptance_ratio"]) < TOL.RATIO
        assert np.array_equal(ob["zhist"][r], ref["zhist"])
    eng.close()


def test_thermalisation_and_E0_restart(S, O):
    """SMC.c:110-125 (2A thermalisation) and the E[0] restart of the production series (:194)"""
    R0 = O.fcc(4, 4)
    s = O.make_sys(256)
    for flags, restart in ((S.FLAGS_REFERENCE, True), (S.FLAG_WALLS, False)):
        eng, p = make_engine(S, O, R0, 2, flags=flags | S.FLAG_SERIES)
        eng.run(4, 6, 2)
        ob = eng.observables()
        Es, jj = eng.series(6)
        ta = eng.therm_acceptance()
        for r in range(2):
            ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 4, 6, 2, e0_restart=restart)
            if restart:                                      # entry 0 is the energy of the START (SMC.c:194)
                TOL.assert_energy(Es[r][0], ref["E"][0], 0)
            TOL.assert_series(Es[r], ref["E"], k0=4, what="replica %d" % r)     # production entry j: 4 + j sweeps from the start
            TOL.assert_mean_energy(ob["meanE"][r], ref["meanE"], 10)
            assert abs(ta[r] - ref["therm_acceptance"]) < TOL.RATIO
            assert np.array_equal(jj[r], ref["jj"])
            assert np.array_equal(ob["zhist"][r], ref["zhist"]) and ref["gathers"] == 3
        eng.close()


def test_thermalisation_at_the_benchmark_geometry(S, O):
    """the same through sweep_kernel_mc64 (N=4096): 2 thermalisation sweeps at 2A, then 3 production sweeps with a
    gather every 2 -- a launch group that ends mid-chunk, the E[0] restart, per-replica positions and explicit seeds"""
    base = O.fcc(8, 16)
    R0 = np.stack([base, np.roll(base.reshape(-1, 3), 11, axis=0).ravel()])
    seeds = np.array([777, 2 ** 31 + 5], dtype=np.uint32)
    p = S.default_params(4096, 2, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=64, tune_waves=1)
    with S.Engine(p) as eng:
        assert eng.kernel_form[1] == "smcx::sweep_kernel_mc64"
        eng.upload(R0, O.W_FIXTURE, seeds)
        eng.run(2, 3, 2)
        ob = eng.observables()
        Es, jj = eng.series(3)
        ta = eng.therm_acceptance()
    s = sys_of(O, p)
    for r in range(2):
        ref = O.chain(s, int(seeds[r]), R0[r], O.W_FIXTURE, T, A, 2, 3, 2, e0_restart=True)
        assert np.array_equal(jj[r], ref["jj"]) and ref["jj"].sum() > 0
        TOL.assert_series(Es[r], ref["E"], k0=2, what="replica %d" % r)
        assert abs(ta[r] - ref["therm_acceptance"]) < TOL.RATIO
        TOL.assert_mean_energy(ob["meanE"][r], ref["meanE"], 5)
        assert np.array_equal(ob["zhist"][r], ref["zhist"])


def test_explicit_seeds_and_per_replica_positions(S, O):
    rs = np.random.RandomState(2)
    base = O.fcc(4, 4)
    R0 = np.stack([base, base + 0.0, base])
    R0[1] = np.roll(base.reshape(-1, 3), 7, axis=0).ravel()
    seeds = np.array([99, 12345, 2 ** 31 + 17], dtype=np.uint32)
    p = S.default_params(256, 3)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE, seeds)
    eng.run(0, 3, 1)
    ob = eng.observables()
    s = sys_of(O, p)
    for r in range(3):
        ref = O.chain(s, int(seeds[r]), R0[r], O.W_FIXTURE, T, A, 0, 3, 1)
        assert int(ob["accepted"][r]) == ref["accepted"]
        TOL.assert_energy(ob["E_last"][r], ref["Efinal"], 3)
    eng.close()


def test_zero_uniform_follows_exp_underflow(S, O):
    """rand() == 0 as the acceptance uniform (SMC.c:335): the reference accepts while exp(-x/T) > 0
    and rejects once it underflows to zero (x/T >= 1075 ln 2); the kernels compare log u < -x/T, so
    u = 0 must be stored as that edge, not as log 0 = -inf.  Two particles at the Lennard-Jones
    minimum, T = A = 1e-3, RNG states crafted so that both acceptance uniforms of the sweep (draws
    7 and 8 of 4N+1 = 9) are exactly 0; the 256 replicas cover x/T above and just below the edge."""
    nrep, N = 256, 2
    R0 = np.array([0.0, 0.0, 0.0, 1.1225, 0.0, 0.0])
    st = np.stack([S.rng_seed(1000 + r) for r in range(nrep)])
    for r in range(nrep):   # new[i] = old[i] + old[i-3] + old[i-6] + old[i+22]: make new[7] = new[8] = 0
        o = [int(v) for v in st[r, :31]]
        st[r, 7] = (-(o[4] + o[1] + o[29])) & 0xFFFFFFFF
        st[r, 8] = (-(o[5] + o[2] + o[30])) & 0xFFFFFFFF
    p = S.default_params(N, nrep, T=1e-3, A=1e-3)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    eng.rng_import(st)
    eng.run(0, 1, 1)
    ob = eng.observables()
    Rg = eng.positions()
    eng.close()
    s = sys_of(O, p)
    E0 = O.total_energy(s, R0, O.W_FIXTURE)
    kinds = {"underflow": 0, "tiny": 0, "plain": 0}
    for r in range(nrep):
        R = R0.copy()
        acc, E1, tr = O.sweep(s, O.Rng.from_state(st[r]), R, O.W_FIXTURE, 1e-3, 1e-3, E=E0, trace=True)
        assert np.all(tr["u"] == 0.0)
        for m in tr:
            kinds["underflow" if m["ap"] == 0.0 else "tiny" if m["ap"] < 1e-200 else "plain"] += 1
        assert int(ob["accepted"][r]) == acc, (r, tr["ap"], tr["accepted"])
        TOL.assert_energy(ob["E_last"][r], E1, 1)
        TOL.assert_positions(Rg[r], R, 1)
    assert kinds["underflow"] > 100 and kinds["tiny"] > 10 and kinds["plain"] > 50, kinds


# ------------------------------------------------------------------ invariants at any size
def test_determinism_sharding_and_resume(S, O):
    """bit-identical results for: a repeated run; a replica computed inside a different
    shard (seeds follow the global replica index); 2+3 sweeps vs 5 sweeps with the RNG
    state exported and re-imported in between (checkpoint/resume of main.c:98-108, 162-170)."""
    R0 = O.fcc(8, 4)
    eng, p = make_engine(S, O, R0, 8)
    eng.run(0, 5, 1)
    a = eng.observables(); Ra = eng.positions()
    eng.close()
    eng, _ = make_engine(S, O, R0, 8)
    eng.run(0, 5, 1)
    b = eng.observables()
    assert np.array_equal(a["E_last"], b["E_last"]) and np.array_equal(Ra, eng.positions())
    eng.close()
    # shard [5, 8) of the same ensemble, different geometry of the launch
    eng, _ = make_engine(S, O, R0, 3, first_replica=5)
    eng.run(0, 5, 1)
    c = eng.observables()
    assert np.array_equal(c["E_last"], a["E_last"][5:]) and np.array_equal(c["accepted"], a["accepted"][5:])
    eng.close()
    # resume
    eng, _ = make_engine(S, O, R0, 8)
    eng.run(0, 2, 1)
    st, R2 = eng.rng_export(), eng.positions()
    eng.close()
    eng2 = S.Engine(S.default_params(1024, 8))
    eng2.upload(R2, O.W_FIXTURE)
    eng2.rng_import(st)
    eng2.run(0, 3, 1)
    assert np.array_equal(eng2.positions(), Ra)
    eng2.close()


def test_all_geometries_agree(S, O):
    """every (slots, waves) instantiation that fits N=1024 gives the same chain to rounding"""
    R0 = O.fcc(8, 4)
    ref = O.chain(O.make_sys(1024), 12345, R0, O.W_FIXTURE, T, A, 0, 3, 1)
    for slots, waves in ((16, 1), (32, 1), (64, 1), (16, 2), (32, 2), (16, 4), (32, 4), (16, 8), (32, 8),
                         (16, 16), (32, 16)):
        eng, p = make_engine(S, O, R0, 2, tune_slots=slots, tune_waves=waves)
        assert eng.geometry[:2] == (slots, waves)
        eng.run(0, 3, 1)
        ob = eng.observables()
        assert int(ob["accepted"][0]) == ref["accepted"], (slots, waves)
        TOL.assert_energy(ob["E_last"][0], ref["Efinal"], 3, "%d x %d" % (slots, waves))
        TOL.assert_positions(eng.positions()[0], ref["R"], 3, "%d x %d" % (slots, waves))
        eng.close()


def test_full_size_invariants_N4096(S, O):
    """BASELINE config 3 shape (N=4096; replicas reduced to keep the test short): the
    incremental energy equals a from-scratch recomputation, histograms conserve particles,
    the first replicas match the oracle."""
    R0 = O.fcc(8, 16)
    nrep = 256
    eng, p = make_engine(S, O, R0, nrep)
    eng.run(1, 3, 2)
    ob = eng.observables()
    g, oob = eng.hist_info()
    Erec = eng.total_energy()
    # thermalisation changed the energy but the series restarted from E[0] (SMC.c:194):
    # compare energy differences instead
    eng2, _ = make_engine(S, O, R0, nrep, flags=S.FLAG_WALLS)
    eng2.run(1, 3, 2)
    ob2 = eng2.observables()
    assert np.all(rel(ob2["E_last"], eng2.total_energy(), 1.0) < TOL.INCREMENTAL)
    assert np.array_equal(ob2["accepted"], ob["accepted"])
    assert np.all(g == 1) and np.all(oob == 0) and np.all(ob["zhist"].sum(axis=1) == 4096)
    assert np.all(np.isfinite(Erec))
    s = sys_of(O, p)
    ref = O.chain(s, 12345, R0, O.W_FIXTURE, T, A, 1, 3, 2)
    assert int(ob["accepted"][0]) == ref["accepted"]
    TOL.assert_mean_energy(ob["meanE"][0], ref["meanE"], 4)
    # distinct seeds really give distinct chains
    assert len(np.unique(ob["E_last"])) > nrep // 2
    eng.close(); eng2.close()


def test_edge_cases(S, O):
    # the smallest system: two particles, no neighbours inside the cutoff
    R0 = np.array([0.0, 0.0, -10.0, 5.0, 5.0, 10.0])
    p = S.default_params(2, 1)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    eng.run(0, 4, 2)
    ob = eng.observables()
    ref = O.chain(sys_of(O, p), 12345, R0, O.W_FIXTURE, T, A, 0, 4, 2)
    assert int(ob["accepted"][0]) == ref["accepted"]
    TOL.assert_energy(ob["E_last"][0], ref["Efinal"], 4)
    assert np.array_equal(ob["zhist"][0], ref["zhist"])
    eng.close()
    # zero sweeps: observables are those of the initial state
    eng, p = make_engine(S, O, O.fcc(4, 4), 2)
    eng.run(0, 0, 1)
    ob = eng.observables()
    assert np.all(ob["accepted"] == 0) and np.all(ob["zhist"] == 0)
    with pytest.raises(S.SmcxError):
        eng.run(0, 1, 0)  # gather_lapse < 1
    eng.close()
    # run before upload
    eng = S.Engine(S.default_params(256, 1))
    with pytest.raises(S.SmcxError) as e:
        eng.run(0, 1, 1)
    assert e.value.status == S.ERR_STATE
    eng.close()


def test_observable_export_to_device_memory(S, O):
    """the packed block the RCCL gather moves (smcx_export_observables_device)"""
    import sys, os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "montecarlo-surfacer_amd"))
    import dist as D
    eng, p = make_engine(S, O, O.fcc(4, 4), 5)
    eng.run(0, 4, 2)
    ob = eng.observables()
    nbytes = eng.obs_device_bytes()
    buf = torch.zeros(nbytes // 8, dtype=torch.float64, device="cuda")
    eng.export_observables_device(buf.data_ptr(), nbytes)
    torch.cuda.synchronize()
    got = D.gather_observables(buf, 5, p.Ncz)
    assert np.array_equal(got["accepted"], ob["accepted"].astype(np.float64))
    assert np.array_equal(got["zhist"], ob["zhist"].astype(np.float64))
    sm = D.summarise(got, p.N, 4)
    assert np.allclose(sm["meanE"], ob["meanE"], rtol=TOL.EXACT_SERIES, atol=0)
    assert np.allclose(sm["acceptance_ratio"], ob["acceptance_ratio"], rtol=TOL.EXACT_SERIES, atol=0)
    eng.close()


# ------------------------------------------------------------------ SURVEY 8f "next" rows
def test_full_density_mobility_and_pressure(S, O):
    """8f.1/8f.2: the Ncx x Ncx x Ncz occupancy D and mobility Mu of localDensityAndMobility
    (SMC.c:912-927) and pressure + wallsPressure of every gather (SMC.c:140, 696-720, 862-895)"""
    R0 = dense_state(O); R0[3 * 82 + 2] = 119.0; R0[3 * 83 + 2] = -119.5
    nrep = 3
    flags = S.FLAGS_REFERENCE | S.FLAG_FULL_HIST | S.FLAG_PRESSURE | S.FLAG_SERIES
    eng, p = make_engine(S, O, R0, nrep, flags=flags)
    eng.run(1, 6, 2)
    D, Mu = eng.density()
    P = eng.pressure_series()
    ob = eng.observables()
    s = sys_of(O, p)
    assert P.shape == (nrep, 3)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 1, 6, 2, full_hist=True, pressure=True)
        assert np.array_equal(D[r], ref["D"]) and np.array_equal(Mu[r], ref["Mu"])
        assert D[r].sum() == 3 * 1024 and np.array_equal(D[r].reshape(33, 33, 33).sum(axis=(0, 1)), ob["zhist"][r])
        TOL.assert_virial(P[r], ref["P"], 1 + 2 * np.arange(1, 4) - 1, "replica %d" % r)   # gather g before sweep 2 g - 1 of production
    eng.close()


def test_host_sMC_driver(S, O):
    """8f.1: smcx_host_sMC, the C driver standing for sMC (SMC.c:21-267): ensemble results against
    the oracle chain of every replica -- energy, acceptance, pressure with the reference's P[] indexing
    (SMC.c:138-140, 207-208, 246-247), tau and cv (SMC.c:234-235, 249-250), cluster counts per analysis"""
    L = 8.0
    R0 = film_state(O, 4, 4, L, 0.05, 5)
    nrep, maxsteps, gl, eq = 3, 8, 2, 1
    flags = S.FLAGS_REFERENCE | S.FLAG_SERIES | S.FLAG_PRESSURE | S.FLAG_CLUSTERS
    p = S.default_params(256, nrep, L=L, flags=flags, lca_time=2)
    sim = S.host_sMC(p, O.W_FIXTURE, R0, maxsteps, gl, eq)
    s = sys_of(O, p)
    rho = 256 / (L * L * 240.0)
    E, acc, Pm, dP, tau, cv, l1 = [], [], [], [], [], [], []
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, eq, maxsteps, gl, pressure=True, lca_time=2)
        E.append(ref["meanE"]); acc.append(ref["acceptance_ratio"])
        gs = maxsteps // gl
        Pk = np.array([rho * T] + [ref["P"][k - 1] + rho * T for k in range(1, gs)])
        Pm.append(Pk.mean()); dP.append(np.sqrt(Pk.var()))
        a = O.fft_acf(ref["E"] + 3 * 256 * T / 2, 2500000)
        tau.append(a.sum()); cv.append(np.var(ref["E"]) / T ** 2)
        l1.append(ref["lca"]["n1"] / ref["lca"]["analyses"])
        TOL.assert_mean_energy(sim["rep_E"][r], ref["meanE"], eq + maxsteps)
        assert rel(sim["rep_acceptance"][r], ref["acceptance_ratio"]) < TOL.RATIO
        TOL.assert_positions(sim["Rfinal"][r], ref["R"], eq + maxsteps)
    TOL.assert_mean_energy(sim["E"], np.mean(E), eq + maxsteps)
    assert rel(sim["acceptance_ratio"], np.mean(acc)) < TOL.RATIO
    TOL.assert_virial(sim["P"], np.mean(Pm), eq + maxsteps)
    assert rel(sim["dP"], np.mean(dP), scale=TOL.EXACT_SERIES) < TOL.OBSERVABLE
    assert rel(sim["tau"], np.mean(tau)) < TOL.OBSERVABLE and rel(sim["cv"], np.mean(cv)) < TOL.OBSERVABLE
    assert sim["lca_analyses"] == 2 and rel(sim["l1"], np.mean(l1)) < TOL.RATIO
    assert abs(sim["l2"].sum() - sim["l1"]) < TOL.EXACT_SERIES * max(1.0, sim["l1"])


def test_multi_gpu_c_host_gathers_through_rccl(S, O):
    """smcx_host_sMC_multi (plain C: one handle and one host thread per device, ncclCommInitAll + ncclAllGather of the
    packed observable records -- the one exchange of the path) on the devices this box has.  Against
    smcx_host_sMC (host-side reads of one handle), against the oracle, and against the Python gather of dist.py;
    with two "devices" that are the same GPU (device list [0, 0] is refused by RCCL, so that leg uses the
    host-concatenation fallback) the shard boundaries and the global seeds are exercised."""
    L = 8.0
    R0 = film_state(O, 4, 4, L, 0.05, 5)
    nrep, maxsteps, gl, eq = 5, 8, 2, 1
    flags = S.FLAGS_REFERENCE | S.FLAG_SERIES | S.FLAG_PRESSURE | S.FLAG_CLUSTERS
    p = S.default_params(256, nrep, L=L, flags=flags, lca_time=1)
    one = S.host_sMC(p, O.W_FIXTURE, R0, maxsteps, gl, eq)
    multi = S.host_sMC(p, O.W_FIXTURE, R0, maxsteps, gl, eq, gpus=1)            # RCCL path, one-rank communicator
    for k in ("E", "dE", "acceptance_ratio", "therm_acceptance", "P", "dP", "tau", "cv", "l1"):
        assert multi[k] == one[k] or rel(multi[k], one[k]) < TOL.EXACT_SERIES, k
    for k in ("rep_E", "rep_acceptance", "zprofile", "Rfinal"):
        assert np.array_equal(multi[k], one[k]), k
    for k in ("l2", "l3"):
        assert np.allclose(multi[k], one[k], rtol=TOL.EXACT_SERIES, atol=0), k
    s = sys_of(O, p)
    for r in (0, nrep - 1):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, eq, maxsteps, gl)
        TOL.assert_mean_energy(multi["rep_E"][r], ref["meanE"], eq + maxsteps)
        assert rel(multi["rep_acceptance"][r], ref["acceptance_ratio"]) < TOL.RATIO
    # two shards (3 + 2 replicas) on the same GPU through the host-concatenation fallback: same numbers, global order
    os.environ["SMCX_HOST_GATHER"] = "host"
    try:
        two = S.host_sMC(p, O.W_FIXTURE, R0, maxsteps, gl, eq, gpus=2, devices=[0, 0])
    finally:
        del os.environ["SMCX_HOST_GATHER"]
    for k in ("rep_E", "rep_acceptance", "Rfinal"):
        assert np.array_equal(two[k], one[k]), k
    assert rel(two["E"], one["E"]) < TOL.EXACT_SERIES and np.allclose(two["zprofile"], one["zprofile"], rtol=TOL.EXACT_SERIES, atol=0)
    assert rel(two["P"], one["P"]) < TOL.EXACT_SERIES and rel(two["l1"], one["l1"]) < TOL.EXACT_SERIES
    # bad requests: more devices than the box has, more devices than replicas
    with pytest.raises(S.SmcxError) as e:
        S.host_sMC(p, O.W_FIXTURE, R0, maxsteps, gl, eq, gpus=S.device_count() + 1)
    assert e.value.status == S.ERR_PARAM
    with pytest.raises(S.SmcxError):
        S.host_sMC(S.default_params(256, 1, L=L), O.W_FIXTURE, R0, 2, 1, 0, gpus=2, devices=[0, 0])


def test_smcx_main_with_gpus_option(S, O):
    """the C driver program with --gpus 1: the RCCL gather path end to end, same printout as without"""
    exe = os.path.join(os.path.dirname(S.LIB_PATH), "smcx_main")
    a = subprocess.run([exe, "1", "8", "4", "1.1", "256", "3", "4", "4"], capture_output=True, text=True, timeout=600)
    b = subprocess.run([exe, "--gpus", "1", "1", "8", "4", "1.1", "256", "3", "4", "4"], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-800:], b.stderr[-800:])
    pick = lambda t: [ln for ln in t.splitlines() if ln.startswith(("Mean energy", "Mean pressure", "Average acceptance", "z profile"))]
    assert pick(a.stdout) == pick(b.stdout) and len(pick(a.stdout)) == 4
    assert "gathered by RCCL" in b.stdout


def test_csv_outputs_and_restart(S, O, tmp_path):
    """8f.1: data_/local_/last_state_ files in the reference's formats (SMC.c:75-82, 214-225;
    main.c:162-170), and a restart from last_state (main.c:98-108)"""
    R0 = O.fcc(4, 4)
    flags = S.FLAGS_REFERENCE | S.FLAG_FULL_HIST | S.FLAG_PRESSURE | S.FLAG_SERIES
    eng, p = make_engine(S, O, R0, 2, flags=flags)
    eng.run(0, 8, 2)
    S.write_csv(eng, 8, 2, str(tmp_path))
    Rfin = eng.positions()
    eng.close()
    s = sys_of(O, p)
    rho = 256 / (33.0 * 33.0 * 240.0)
    tag = "N256_M3_r%0.4f_T%0.2f" % (rho, 1.1)
    for r in range(2):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, 8, 2, full_hist=True, pressure=True)
        rows = open(tmp_path / ("data_%s_rank%d.csv" % (tag, r))).read().strip().split("\n")
        assert rows[0] == "E, P, jj" and len(rows) == 1 + 4
        for k in range(4):
            e, pk, j = rows[1 + k].split(",")
            eref = ref["E"][2 * k] + 3 * 256 * 1.1 / 2
            assert abs(float(e) - eref) < TOL.printed(9) + TOL.energy(2 * k, eref)          # %0.9lf
            pref = (ref["P"][k - 1] if k >= 1 else 0.0) + rho * 1.1
            assert abs(float(pk) - pref) < TOL.printed(9) + TOL.virial(2 * k, pref)
            assert int(j) == ref["jj"][k]
        loc = np.loadtxt(tmp_path / ("local_%s_rank%d.csv" % (tag, r)), delimiter=",", skiprows=1)
        assert loc.shape == (33 ** 3, 5)
        assert np.array_equal(loc[:, 3].astype(np.uint64), ref["D"]) and np.array_equal(loc[:, 4].astype(np.uint64), ref["Mu"])
        assert np.array_equal(loc[34, :3], [0, 1, 1])  # nx, ny, nz order: k fastest
        last = S.read_last_state(str(tmp_path / ("last_state_%s_rank%d.csv" % (tag, r))), 256)
        assert np.abs(last - Rfin[r]).max() < 3 * TOL.printed(12)  # %0.12f
    # restart replica 0 from its file: the chain continues from the stored (rounded) positions
    eng2 = S.Engine(S.default_params(256, 1))
    eng2.upload(S.read_last_state(str(tmp_path / ("last_state_%s_rank0.csv" % tag)), 256), O.W_FIXTURE)
    eng2.run(0, 1, 1)
    assert eng2.observables()["zhist"][0].sum() == 256
    eng2.close()


def test_energy_autocorrelation(S, O):
    """8f.3: fft_acf of the production energy series (SMC.c:1051-1089, 234-235, 250) via hipFFT"""
    R0 = O.fcc(4, 4)
    for maxsteps in (40, 33):  # even and odd series length
        eng, p = make_engine(S, O, R0, 3, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES)
        eng.run(0, maxsteps, 10)
        Es, _ = eng.series(maxsteps)
        acf, tau, cv = eng.acf()
        ob = eng.observables()
        k = (maxsteps + 1) // 2 - 2
        assert acf.shape == (3, k)
        for r in range(3):
            ref = O.fft_acf(Es[r], 2500000)   # same series in, restated transform out
            assert np.abs(acf[r] - ref).max() < TOL.FFT
            assert abs(tau[r] - ref.sum()) < TOL.FFT * max(1.0, abs(ref.sum()))
            assert abs(cv[r] - ob["dE"][r] ** 2 / 1.1 ** 2) < TOL.EXACT_SERIES * max(1.0, cv[r])
        acf5, _, _ = eng.acf(k_max=5)
        assert acf5.shape == (3, 5) and np.abs(acf5 - acf[:, :5]).max() < TOL.EXACT_SERIES
        eng.close()


@pytest.mark.parametrize("N,lat,L,slots,waves,nsw,walls", [
    (4096, (8, 16), 33.0, 64, 1, 3, True),    # benchmark lattice, one wavefront per replica
    (4096, (16, 4), 33.0, 32, 2, 2, True),    # dense film: ~40 neighbours inside the cutoff per particle
    (4096, (16, 4), 33.0, 16, 4, 2, True),
    (1024, (8, 4), 33.0, 16, 1, 6, True),
    (1024, (8, 4), 16.0, 16, 1, 3, True),     # box of 5.3 cutoffs: most probes within a cutoff of a periodic edge
    (2000, (8, 8), 33.0, 32, 1, 3, True),     # ragged: N is not a multiple of 64
    (256, (4, 4), 33.0, 0, 0, 8, True),       # small N: the screened kernel asked for by tune_kernel alone (S=16)
    (1024, (8, 4), 33.0, 16, 1, 3, False),    # no walls, a slab of particles beyond the fp32-safe z range
    (4000, (8, 16), 33.0, 64, 1, 2, False),   # z as fp16 in LDS, ragged N (padding slots), no walls, part of the
                                              # system beyond the fp16-safe range: unsafe particles AND unsafe probes
])
def test_screened_kernel_matches_fp64_kernel(S, O, N, lat, L, slots, waves, nsw, walls):
    """the compact-copy (int16 x,y + fp32 z) screen of smcx_sweep_mx.hip only pre-selects pairs:
    every pair inside the cutoff is then decided and evaluated in fp64 exactly as the fp64 kernels
    do.  Two builds of the same fp64 arithmetic differ by rounding (FMA contraction, summation
    order: 1e-14 after a sweep, also between two geometries of the fp64 kernel); one pair missed
    by the screen would shift the energy by >= 4|V(rc)| = 5e-3.  Accept decisions must be the same."""
    rs = np.random.RandomState(N + slots)
    R0 = O.fcc(lat[0], lat[1], L=L).reshape(-1, 3)[:N].copy()
    R0 += 0.05 * rs.standard_normal(R0.shape)
    R0[:, 0] -= L * np.rint(R0[:, 0] / L); R0[:, 1] -= L * np.rint(R0[:, 1] / L)
    R0[:2, 2] = [119.99, -119.999]            # at both walls
    flags = S.FLAGS_REFERENCE | S.FLAG_SERIES
    if not walls:                              # particles beyond zsafe are always candidates, probes there flag all
        flags = S.FLAG_E0_RESTART | S.FLAG_SERIES
        if slots == 64:                        # zsafe = 128 (fp16 z): move the upper part of the film across it
            R0[R0[:, 2] > 6.0, 2] += 105.0
            assert (np.abs(R0[:, 2]) > 128).sum() > 100 and (np.abs(R0[:, 2]) < 128).sum() > 100
        else:                                  # zsafe = 2 Lz (fp32 z)
            R0[R0[:, 2] > 6.0, 2] += 600.0
            assert (R0[:, 2] > 500).sum() > 100
    nrep = 3
    out = []
    for kernel in (1, 2):
        fp64_slots, fp64_waves = (slots, waves) if kernel == 2 or S.geometry_supported_fp64(slots, waves) else (0, 0)
        if slots == 0:
            fp64_slots, fp64_waves = 0, 0
        p = S.default_params(N, nrep, L=L, tune_slots=fp64_slots, tune_waves=fp64_waves, tune_kernel=kernel,
                             flags=flags)
        eng = S.Engine(p)
        assert eng.kernel_form[0] == kernel
        eng.upload(R0.ravel(), O.W_FIXTURE)
        eng.run(1, nsw, 1)
        E, jj = eng.series(nsw)
        out.append((eng.positions().copy(), E.copy(), jj.copy(), eng.observables()["zhist"].copy()))
        eng.close()
    (Ra, Ea, ja, za), (Rb, Eb, jb, zb) = out
    assert np.array_equal(ja, jb) and ja.sum() > 0
    TOL.assert_series(Ea, Eb, k0=1, what="screened against fp64 kernel")      # (one thermalisation sweep in front)
    TOL.assert_positions(Ra, Rb, 1 + nsw)
    assert np.array_equal(za, zb)


@pytest.mark.parametrize("case", ["dense_film", "ragged_no_walls", "thin_film_all_groups"])
def test_byte_screen_kernel_corner_cases_against_fp64_kernel(S, O, case):
    """sweep_kernel_mc64 (z-ordered cells, one word per cell) against the all-fp64 kernels where its rare paths run:
    a dense film (fcc(16,4) in L = 33: ~60 candidates per probe, so fewer lanes without a candidate than wall sites
    -> the fixed-lane fallback for the specials, several evaluation rounds, every group in z reach); no walls with a
    ragged N (empty cells at the end of the z order) and part of the film moved far up (two separated slabs: most
    groups out of reach of every probe; z beyond the int16 of the cells cannot be uploaded in a box this kernel
    serves: |z| <= 4 Lz < 32766 L/256); a film thinner than the
    cutoff (fcc(24,1) in L = 48, the widest box whose L/256 resolves the cutoff in 16 units: all 9 groups in reach of
    every probe, both flag words used)."""
    rs = np.random.RandomState(7)
    flags = S.FLAGS_REFERENCE | S.FLAG_SERIES
    N, L, nsw = 4096, 33.0, 2
    if case == "dense_film":
        R0 = O.fcc(16, 4, L=L).reshape(-1, 3).copy()
    elif case == "ragged_no_walls":
        N = 4000
        R0 = O.fcc(8, 16, L=L).reshape(-1, 3)[:N].copy()
        flags = S.FLAG_E0_RESTART | S.FLAG_SERIES
    else:
        N, L = 2304, 48.0
        R0 = O.fcc(24, 1, L=L).reshape(-1, 3).copy()
    R0 += 0.05 * rs.standard_normal(R0.shape)
    R0[:, 0] -= L * np.rint(R0[:, 0] / L); R0[:, 1] -= L * np.rint(R0[:, 1] / L)
    if case == "ragged_no_walls":
        R0[R0[:, 2] > 6.0, 2] += 700.0
        assert (R0[:, 2] > 690).sum() > 100 and (np.abs(R0[:, 2]) < 100).sum() > 100
    out = []
    for kernel, (slots, waves) in ((1, (0, 0)), (2, (64, 1))):
        p = S.default_params(N, 3, L=L, tune_slots=slots, tune_waves=waves, tune_kernel=kernel, flags=flags)
        with S.Engine(p) as eng:
            if kernel == 2:
                assert eng.kernel_form[1] == "smcx::sweep_kernel_mc64", eng.kernel_form
            eng.upload(R0.ravel(), O.W_FIXTURE)
            eng.run(1, nsw, 1)
            E, jj = eng.series(nsw)
            out.append((eng.positions().copy(), E.copy(), jj.copy(), eng.observables()["zhist"].copy()))
    (Ra, Ea, ja, za), (Rb, Eb, jb, zb) = out
    assert np.array_equal(ja, jb) and ja.sum() > 0
    TOL.assert_series(Ea, Eb, k0=1, what="screened against fp64 kernel")      # (one thermalisation sweep in front)
    TOL.assert_positions(Ra, Rb, 1 + nsw)
    assert np.array_equal(za, zb)


@pytest.mark.parametrize("M,N,lat,slots,waves", [(1, 4096, (8, 16), 64, 1), (2, 4096, (8, 16), 64, 1), (5, 4096, (8, 16), 64, 1),
                                                (2, 1024, (8, 4), 16, 1), (5, 2048, (8, 8), 32, 1),
                                                (1, 1024, (8, 4), 16, 1), (5, 1024, (8, 4), 16, 1),   # N <= 1024: ml16
                                                (2, 9216, (12, 16), 64, 8), (5, 9216, (12, 16), 64, 8),   # two teams (M = 5: 26 wall lanes + 2 side lanes in front of the list)
                                                (5, 6144, (16, 6), 32, 4)])                                # 4 wavefronts
def test_wall_grids_other_than_3x3_against_oracle(S, O, M, N, lat, slots, waves):
    """the hand-scheduled kernels give the M^2 wall sites and the plane to the first lanes without a candidate
    (their rank = row of the wall table): M = 1, 2, 5 (2, 5 and 26 special lanes) against the oracle, with the film
    pushed against the lower wall so that the sites act on it (SMC.c:729-813)."""
    rs = np.random.RandomState(M)
    R0 = O.fcc(*lat).reshape(-1, 3).copy()
    R0[:, 2] += -118.5 - R0[:, 2].min()
    W = np.empty(2 * M * M)
    W[0::2] = rs.uniform(800.0, 1050.0, M * M); W[1::2] = rs.uniform(50.0, 62.0, M * M)
    nsw, nrep = 2, 2
    p = S.default_params(N, nrep, M=M, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=slots, tune_waves=waves)
    with S.Engine(p) as eng:
        want = {(64, 8): "mt64x8", (32, 4): "mc32x4", (16, 1): "ml16"}.get((slots, waves), "mc%d" % slots)
        assert eng.kernel_form[1] == "smcx::sweep_kernel_" + want, eng.kernel_form
        eng.upload(R0.ravel(), W)
        eng.run(0, nsw, 1)
        Es, jj = eng.series(nsw)
        Rg = eng.positions()
    s = sys_of(O, p)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0.ravel(), W, T, A, 0, nsw, 1)
        assert np.array_equal(jj[r], ref["jj"]) and ref["jj"].sum() > 0
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        TOL.assert_positions(Rg[r], ref["R"], nsw)
    # the walls matter in this state: without them the first sweep's energy differs
    assert abs(ref["E"][0]) > 0


def test_int16_z_ordered_kernel_unsafe_particles_against_fp64_kernel(S, O):
    """a box whose L/256 does not resolve the cutoff (L = 60) runs sweep_kernel_mb64 (int16 x,y in registers, int16 z
    in LDS, cells in z order).  No walls and the upper part of the film moved beyond the int16 z range (zsafe = 479):
    those particles are always candidates, a proposal there flags all real cells of a ragged N, and the groups'
    z ranges hold clamped values -- against the all-fp64 kernels."""
    rs = np.random.RandomState(11)
    N, L, nsw = 4000, 60.0, 2
    R0 = O.fcc(10, 10, L=L).reshape(-1, 3)[:N].copy()
    R0 += 0.05 * rs.standard_normal(R0.shape)
    R0[:, 0] -= L * np.rint(R0[:, 0] / L); R0[:, 1] -= L * np.rint(R0[:, 1] / L)
    R0[R0[:, 2] > 6.0, 2] += 600.0
    assert (R0[:, 2] > 480).sum() > 100 and (np.abs(R0[:, 2]) < 100).sum() > 100
    out = []
    for kernel, (slots, waves) in ((1, (0, 0)), (2, (64, 1))):
        p = S.default_params(N, 3, L=L, tune_slots=slots, tune_waves=waves, tune_kernel=kernel,
                             flags=S.FLAG_E0_RESTART | S.FLAG_SERIES)
        with S.Engine(p) as eng:
            if kernel == 2:
                assert eng.kernel_form[1] == "smcx::sweep_kernel_mb64", eng.kernel_form
            eng.upload(R0.ravel(), O.W_FIXTURE)
            eng.run(1, nsw, 1)
            E, jj = eng.series(nsw)
            out.append((eng.positions().copy(), E.copy(), jj.copy()))
    (Ra, Ea, ja), (Rb, Eb, jb) = out
    assert np.array_equal(ja, jb) and ja.sum() > 0
    TOL.assert_series(Ea, Eb, k0=1, what="mb64 against fp64 kernel")
    TOL.assert_positions(Ra, Rb, 1 + nsw)


def film_state(O, Na, Nz, L, jitter, seed):
    """a dense fcc film (nearest neighbours inside LCA_cutoff) with thermal jitter"""
    rs = np.random.RandomState(seed)
    R = O.fcc(Na, Nz, L=L).reshape(-1, 3).copy()
    R += jitter * rs.standard_normal(R.shape)
    R[:, 0] -= L * np.rint(R[:, 0] / L)
    R[:, 1] -= L * np.rint(R[:, 1] / L)
    return R.ravel()


@pytest.mark.parametrize("N,Na,Nz,L,cut", [(256, 4, 4, 8.0, 1.7), (1024, 8, 4, 16.5, 1.7),
                                           (256, 4, 4, 6.6, 1.9), (108, 3, 3, 6.0, 1.7)])
def test_cluster_analysis_matches_oracle(S, O, N, Na, Nz, L, cut):
    """8f.4: clusterAnalysis (SMC.c:971-1045) -- num1/num2/num3 of every pair entry, bit-exact,
    with the reference's overlapping pair index; the (6.6, 1.9) case has more than 8 common
    neighbours per pair (the reference's common_nn[8] overflow, counted identically)"""
    nrep = 3
    R0 = np.stack([film_state(O, Na, Nz, L, 0.08, 10 * N + r) for r in range(nrep)])
    p = S.default_params(N, nrep, L=L, lca_cutoff=cut)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    eng.cluster_update()
    eng.cluster_update()
    n1, h2, h3, ov, k = eng.cluster_counts()
    assert k == 2
    for r in range(nrep):
        ref, ovr = O.cluster_analysis(N, R0[r], L, cut)
        got, ovg = eng.cluster_analysis(r)
        assert np.array_equal(got, ref), np.argwhere(got != ref)[:5]
        assert ovg == ovr
        c1, c2, c3 = O.cluster_counts(N, ref)
        assert c1 > N and n1[r] == 2 * c1
        assert np.array_equal(h2[r], 2 * c2) and np.array_equal(h3[r], 2 * c3) and ov[r] == 2 * ovr
        if cut > 1.8:
            assert ovr > 0
    # entries shared by (l, l-1) and (l+1, 0): present in the comparison above; spot-check one
    eng.close()


def test_cluster_analysis_during_run(S, O):
    """8f.4: the analysis at the gathers k with k % LCA_TIME == 0 (SMC.c:143) inside smcx_run"""
    L = 8.0
    R0 = film_state(O, 4, 4, L, 0.05, 77)
    nrep = 2
    flags = S.FLAGS_REFERENCE | S.FLAG_CLUSTERS
    p = S.default_params(256, nrep, L=L, flags=flags, lca_time=2)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    eng.run(1, 9, 2)   # gathers k = 1..4, analyses at k = 2, 4
    n1, h2, h3, ov, k = eng.cluster_counts()
    assert k == 2
    s = sys_of(O, p)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 1, 9, 2, lca_time=2, lca_cutoff=1.7)["lca"]
        assert ref["analyses"] == 2
        assert n1[r] == ref["n1"] and np.array_equal(h2[r], ref["h2"]) and np.array_equal(h3[r], ref["h3"])
        assert ov[r] == ref["overflow"]
    eng.run(0, 2, 1)   # counters restart with every run (SMC.c:58-60)
    assert eng.cluster_counts()[4] == 1
    eng.close()


@pytest.mark.parametrize("N", [66, 130, 1000, 2050, 4000, 5000, 6144, 8192, 9000])
def test_ragged_sizes_with_padding(S, O, N):
    """N that does not fill the launch geometry (padding lanes/slots, partial last slot, runs
    that wrap inside a slot): two free-running sweeps against the oracle.  4096 < N <= 8192 runs the z-ordered
    four-wavefront kernel sweep_kernel_mc32x4 (round 3: that range used to fall back to sweep_kernel_mx)."""
    rs = np.random.RandomState(N)
    Na = int(np.ceil((N / 4.0) ** (1 / 3.0))) + 1
    R0 = O.fcc(Na, Na).reshape(-1, 3)
    R0 = R0[rs.permutation(len(R0))[:N]].ravel().copy()
    nrep = 3 if N <= 4096 else 2
    eng, p = make_engine(S, O, R0, nrep)
    assert eng.geometry[0] * eng.geometry[1] * 64 >= N
    if 4096 < N <= 8192:
        assert eng.kernel_form[1] == "smcx::sweep_kernel_mc32x4" and eng.geometry[:2] == (32, 4)
    eng.run(0, 2, 1)
    ob = eng.observables()
    Rg = eng.positions()
    s = sys_of(O, p)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, 2, 1)
        assert int(ob["accepted"][r]) == ref["accepted"], (N, r)
        TOL.assert_energy(ob["E_last"][r], ref["Efinal"], 2, "N=%d replica %d" % (N, r))
        TOL.assert_positions(Rg[r], ref["R"], 2, "N=%d replica %d" % (N, r))
        assert np.array_equal(ob["zhist"][r], ref["zhist"])
    eng.close()


@pytest.mark.parametrize("N,lat,kw,kernel", [
    (4096, (8, 16), dict(cutoff=2.5, T=0.9, A=0.5, a0=2.0e-8, b0=4.0e-5, Ncx=20, Ncz=50), "mc64"),
    (1024, (8, 4), dict(cutoff=3.5, T=1.4, A=2.0, Ncx=33, Ncz=17), "ml16"),
    (16384, (16, 16), dict(cutoff=2.2, T=2.0, A=0.05, a0=1.0e-8, b0=1.0e-5, tune_slots=64, tune_waves=8), "mt64x8"),
    (6144, (16, 6), dict(cutoff=2.6, T=1.0, A=0.3), "mc32x4"),
    (256, (4, 4), dict(cutoff=4.0, T=0.7, A=1.5, L=20.0, Lz=100.0, Ncx=12, Ncz=40), None),
])
def test_runtime_parameters_other_than_the_references_macros(S, O, N, lat, kw, kernel):
    """What the reference fixes at compile time (SMC.h:26-58: LJ_CUTOFF, a0, b0, Ncx, Ncz; main.c:18, 48-51: T, gamma) are run-time
    fields of smcx_params.  Every other test runs the reference's values; here cutoff, temperature, step size, plane
    coefficients, cell counts and (last case) the box differ: the screens' thresholds, the groups' reach, the near-wall bound and
    the histogram must all follow.  Two sweeps with a gather before each against the oracle built with the same numbers."""
    rs = np.random.RandomState(N)
    L, Lz = kw.get("L", 33.0), kw.get("Lz", 240.0)
    R0 = O.fcc(lat[0], lat[1], L=L, Lz=Lz).reshape(-1, 3).copy()
    R0 += 0.05 * rs.standard_normal(R0.shape)
    R0[:, 2] += -(Lz / 2 - 1.3) - R0[:, 2].min()          # the film against the lower wall: sites and plane act
    R0[:, 0] -= L * np.rint(R0[:, 0] / L); R0[:, 1] -= L * np.rint(R0[:, 1] / L)
    R0 = R0.ravel()
    nrep, nsw = 3, 2
    p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, **kw)
    with S.Engine(p) as eng:
        if kernel:
            assert eng.kernel_form[1] == "smcx::sweep_kernel_" + kernel, eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        E0 = eng.total_energy()
        eng.run(0, nsw, 1)
        Es, jj = eng.series(nsw)
        ob = eng.observables()
        Rg = eng.positions()
    s = sys_of(O, p)
    assert s.cutoff == kw["cutoff"] and s.Ncz == p.Ncz
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, p.T, p.A, 0, nsw, 1)
        assert np.array_equal(jj[r], ref["jj"]) and ref["jj"].sum() > 0, (r, jj[r], ref["jj"])
        TOL.assert_energy(E0[r], ref["E"][0], 0)
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        TOL.assert_positions(Rg[r], ref["R"], nsw)
        assert np.array_equal(ob["zhist"][r], ref["zhist"]) and len(ref["zhist"]) == p.Ncz


def test_upload_rejects_unwrapped_positions(S, O):
    R0 = O.fcc(4, 4)
    R0[0] = 20.0  # outside [-L/2, L/2]
    eng = S.Engine(S.default_params(256, 1))
    with pytest.raises(S.SmcxError) as e:
        eng.upload(R0, O.W_FIXTURE)
    assert e.value.status == S.ERR_PARAM
    eng.close()
