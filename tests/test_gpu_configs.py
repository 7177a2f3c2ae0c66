"""GPU tests (-m gpu) of the BASELINE.json configurations at their real sizes, of the screened
kernel's no-miss property on the device, of ensemble statistics beyond the chaos horizon, and of
the multi-rank launch path of bench.py.  Everything goes through the C ABI (libsmcx.so); the CPU
oracle is the checker only.

Tolerances: tests/tolerances.py, the ONE statement of them (integers equal; single evaluations; the schedule in the sweep
index for free-running chains, derived from the measured rounding drift; observables at north_star's 1e-6); statistical
comparisons state theirs.
"""
import importlib.util
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import tolerances as TOL
from tolerances import rel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = A = 1.1


def sys_of(O, p):
    return O.make_sys(p.N, M=p.M, L=p.L, Lz=p.Lz, cutoff=p.cutoff, a0=p.a0, b0=p.b0, Ncx=p.Ncx, Ncz=p.Ncz)


def oracle_chains(O, s, seeds, R0, eq, nsw, gl, workers=16):
    """O.chain for many seeds on the host cores (ctypes releases the GIL)"""
    with ThreadPoolExecutor(min(workers, len(os.sched_getaffinity(0)))) as ex:
        return list(ex.map(lambda sd: O.chain(s, sd, R0, O.W_FIXTURE, T, A, eq, nsw, gl), seeds))


# ------------------------------------------------------------------ BASELINE config 5: N = 16384
@pytest.mark.parametrize("slots,waves,name", [
    (64, 4, "smcx::sweep_kernel_mc64x4"),                # 4 wavefronts x 64 cells per lane, z-ordered, byte screen: the
                                                         # geometry rule's choice at this N
    (32, 8, "smcx::sweep_kernel_mc32x8"),                # the same with 8 wavefronts x 32 cells
    (64, 8, "smcx::sweep_kernel_mt64x8"),                # two teams of 4 wavefronts x 64 cells (round 3): the rule's choice
                                                         # up to 256 replicas per GPU
])
def test_config5_N16384_against_oracle(S, O, slots, waves, name):
    """BASELINE configs[4]: N=16384 + wall, fcc(16,16) (the reference's own dense lattice),
    several wavefronts per replica.  2 replicas x 2 sweeps against the oracle chain (SMC.c:278-351
    with K1-K4 at that N): accept counts per sweep bit-equal, energies and positions within the schedule of
    tests/tolerances.py, z histogram equal."""
    R0 = O.fcc(16, 16)
    nsw, nrep = 2, 2
    p = S.default_params(16384, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=slots, tune_waves=waves)
    with S.Engine(p) as eng:
        assert eng.kernel_form == (2, name), eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        E0 = eng.total_energy()
        eng.run(0, nsw, 1)
        ob = eng.observables()
        Es, jj = eng.series(nsw)
        Rg = eng.positions()
    s = sys_of(O, p)
    refs = oracle_chains(O, s, [12345 + r for r in range(nrep)], R0, 0, nsw, 1)
    assert abs(E0[0] - -41824.76491) < TOL.printed(5)   # E0 of the real reference at this lattice (as printed there)
    for r, ref in enumerate(refs):
        assert np.array_equal(jj[r], ref["jj"]), (jj[r], ref["jj"])
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        TOL.assert_mean_energy(ob["meanE"][r], ref["meanE"], nsw)
        TOL.assert_positions(Rg[r], ref["R"], nsw, "replica %d" % r)
        assert np.array_equal(ob["zhist"][r], ref["zhist"])


@pytest.mark.parametrize("slots,waves", [(64, 4), (32, 8), (64, 8)])
def test_several_wavefront_kernel_with_many_accepted_moves(S, O, slots, waves):
    """config 5's lattice is a crystal (66 of 16384 moves accepted per sweep).  The same N in the widest box the byte
    screen serves (L = 48: fcc(16,16) at spacing 3, a third of the density) accepts thousands of moves per sweep, so
    the cell writes by the owning wavefront, the widened group ranges and the row caches of all wavefronts are
    exercised: 2 replicas x 2 sweeps against the oracle."""
    L = 48.0
    R0 = O.fcc(16, 16, L=L)
    nsw, nrep = 2, 2
    p = S.default_params(16384, nrep, L=L, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=slots, tune_waves=waves)
    with S.Engine(p) as eng:
        assert eng.kernel_form[1] == {(64, 4): "smcx::sweep_kernel_mc64x4", (32, 8): "smcx::sweep_kernel_mc32x8",
                                      (64, 8): "smcx::sweep_kernel_mt64x8"}[(slots, waves)]
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 1)
        ob = eng.observables()
        Es, jj = eng.series(nsw)
        Rg = eng.positions()
    refs = oracle_chains(O, sys_of(O, p), [12345 + r for r in range(nrep)], R0, 0, nsw, 1)
    for r, ref in enumerate(refs):
        assert np.array_equal(jj[r], ref["jj"]) and ref["jj"].min() > 1000, (jj[r], ref["jj"])
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        TOL.assert_positions(Rg[r], ref["R"], nsw, "replica %d" % r)
        assert np.array_equal(ob["zhist"][r], ref["zhist"])


def test_config5_N16384_x256_invariants(S, O):
    """config 5 per GPU (2048 replicas / 8 GPUs = 256): the geometry rule's own choice, three sweeps:
    incremental energy = recomputed energy, histograms conserve particles, replica 0 = the oracle's."""
    R0 = O.fcc(16, 16)
    nrep = 256
    p = S.default_params(16384, nrep, flags=S.FLAG_WALLS)
    with S.Engine(p) as eng:
        form, name = eng.kernel_form
        # (the two-team kernel: both teams of four wavefronts hold all cells)
        assert form == 2 and name == "smcx::sweep_kernel_mt64x8" and eng.geometry[:2] == (64, 8), (name, eng.geometry)
        eng.upload(R0, O.W_FIXTURE)
        eng.run(1, 2, 1)
        ob = eng.observables()
        g, oob = eng.hist_info()
        Erec = eng.total_energy()
    assert np.all(rel(ob["E_last"], Erec, 1.0) < TOL.INCREMENTAL)
    assert np.all(g == 2) and np.all(oob == 0) and np.all(ob["zhist"].sum(axis=1) == 2 * 16384)
    assert len(np.unique(ob["accepted"])) > 8          # distinct seeds, distinct chains
    # sixteen replicas spread evenly over the launch (round 5: as many as the oracle affords -- 3 sweeps of N = 16384 are ~10 s of
    # a host core each -- instead of four picked by hand; all eight wavefronts of a workgroup resident beside those of other
    # replicas) against the oracle chain (SMC.c:110-118 thermalisation at 2A, :134-195 production)
    pick = [int(x) for x in np.linspace(0, nrep - 1, 16).round()]
    with ThreadPoolExecutor(min(16, len(os.sched_getaffinity(0)))) as ex:
        refs = list(ex.map(lambda r: O.chain(sys_of(O, p), 12345 + r, R0, O.W_FIXTURE, T, A, 1, 2, 1, e0_restart=False), pick))
    for r, ref in zip(pick, refs):
        assert int(ob["accepted"][r]) == ref["accepted"], r
        TOL.assert_energy(ob["E_last"][r], ref["Efinal"], 3, "replica %d" % r)
        assert np.array_equal(ob["zhist"][r], ref["zhist"]), r


# ------------------------------------------------------------------ BASELINE config 3: N = 4096 x 4096 (the headline)
# 64 replicas spread evenly over the launch (round 5: every replica the oracle affords in ~10 s on the box's 16 cores instead of
# eight picked by hand; first, second, last-but-one and last included)
CONFIG3_PICK = sorted(set([0, 1, 4094, 4095] + [int(x) for x in np.linspace(0, 4095, 62).round()]))


def test_config3_N4096_x4096_headline_launch_against_oracle(S, O):
    """BASELINE configs[2] exactly as bench.py runs it: default parameters, 4096 replicas of N=4096 on one GPU (four
    wavefronts on every SIMD of the chip, the issue-priority table live, `Rs` cycling through the L2s), gather_lapse 10,
    kernel sweep_kernel_mc64.  64 replicas spread over the launch against the oracle chain (SMC.c:278-351 sweeps
    inside sMC's loop, SMC.c:134-195), three sweeps: accepted counts bit-equal, energies and positions within the schedule
    of tests/tolerances.py (every replica, and the median replica); then the same launch with a gather before every sweep
    for the per-sweep series and the z histogram (SMC.c:912-927)."""
    R0 = O.fcc(8, 16)
    nrep, nsw, pick = 4096, 3, CONFIG3_PICK
    p = S.default_params(4096, nrep)                     # bench.py's parameters
    with S.Engine(p) as eng:
        assert eng.kernel_form == (2, "smcx::sweep_kernel_mc64"), eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 10)                              # bench.py: eng.run(0, steps, gather_lapse = 10)
        ob = eng.observables()
        Rg = eng.positions()[pick].copy()
        Erec = eng.total_energy()
    s = sys_of(O, p)
    # one oracle pass serves both launches: the chain does not depend on gather_lapse, only the histogram does (gathered
    # before every sweep here; with gather_lapse 10 no gather falls into three sweeps)
    refs = oracle_chains(O, s, [12345 + r for r in pick], R0, 0, nsw, 1)
    assert np.all(rel(ob["E_last"], Erec, 1.0) < TOL.INCREMENTAL)   # every replica: incremental energy = recomputed energy
    assert len(np.unique(ob["E_last"])) > nrep // 2      # distinct seeds, distinct chains
    for k, (r, ref) in enumerate(zip(pick, refs)):
        assert int(ob["accepted"][r]) == ref["accepted"], r
        TOL.assert_energy(ob["E_last"][r], ref["Efinal"], nsw, "replica %d" % r)
        TOL.assert_mean_energy(ob["meanE"][r], ref["meanE"], nsw, "replica %d" % r)
        assert abs(ob["acceptance_ratio"][r] - ref["acceptance_ratio"]) < TOL.RATIO, r
        assert ob["zhist"][r].sum() == 0
    TOL.assert_positions(Rg, np.stack([ref["R"] for ref in refs]), nsw, "headline launch")
    # the same full launch with a gather before every sweep: per-sweep series and the wall-normal profile
    p = S.default_params(4096, nrep, flags=p.flags | S.FLAG_SERIES)
    with S.Engine(p) as eng:
        assert eng.kernel_form == (2, "smcx::sweep_kernel_mc64"), eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 1)
        ob = eng.observables()
        Es, jj = eng.series(nsw)
        g, oob = eng.hist_info()
    assert np.all(g == nsw) and np.all(oob == 0) and np.all(ob["zhist"].sum(axis=1) == nsw * 4096)
    for r, ref in zip(pick, refs):
        assert np.array_equal(jj[r], ref["jj"]), (r, jj[r], ref["jj"])
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        assert np.array_equal(ob["zhist"][r], ref["zhist"]), r


def test_config3_full_occupancy_from_equilibrating_per_replica_states(S, O):
    """Round 5 (VERDICT r4 "what's weak" #3, next #1b): every other 4096-replica test starts all replicas from ONE fcc lattice and
    runs <= 3 sweeps -- the z sort, the group ranges and the hand-over lists at full occupancy had only ever seen 4096 copies of
    a relaxing crystal.  Here the headline launch runs 500 sweeps (the slab has spread from 66 to ~150 in z, acceptance 0.43 ->
    0.51, every replica its own disordered state: profiles/r05_equil_config3.txt), all positions are downloaded and
    uploaded again per replica (`r0_per_replica`) with fresh explicit seeds, and two further sweeps with a gather before each run
    at full occupancy.  64 replicas spread over the launch are compared with the oracle chain started from THAT replica's
    downloaded state (SMC.c:278-351 inside sMC's loop, SMC.c:134-195): accepted counts and z histograms equal, energies and
    positions within the schedule of tests/tolerances.py.  Also: after the 500 sweeps the energy carried incrementally
    (SMC.c:340-341) equals the energy recomputed from the positions, for every replica."""
    R0 = O.fcc(8, 16)
    nrep, long_run, nsw, pick = 4096, 500, 2, CONFIG3_PICK
    p = S.default_params(4096, nrep)
    with S.Engine(p) as eng:
        assert eng.kernel_form == (2, "smcx::sweep_kernel_mc64"), eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, long_run, 10)
        ob = eng.observables()
        Erec = eng.total_energy()
        Rall = eng.positions()
    drift = rel(ob["E_last"], Erec, 1.0)
    print("after %d sweeps: acceptance %.3f, mean E %.1f, max relative |E_incremental - E_recomputed| %.2e" %
          (long_run, ob["acceptance_ratio"].mean(), ob["E_last"].mean(), drift.max()))
    assert np.all(drift < TOL.INCREMENTAL)
    assert 0.35 < ob["acceptance_ratio"].mean() < 0.7 and np.ptp(Rall[:, 2::3]) > 120.0       # it did leave the lattice
    seeds = (900001 + 7 * np.arange(nrep)).astype(np.uint32)
    p2 = S.default_params(4096, nrep, flags=p.flags | S.FLAG_SERIES)
    with S.Engine(p2) as eng:
        assert eng.kernel_form == (2, "smcx::sweep_kernel_mc64"), eng.kernel_form
        eng.upload(Rall, O.W_FIXTURE, seeds)
        E0 = eng.total_energy()
        eng.run(0, nsw, 1)
        ob2 = eng.observables()
        Es, jj = eng.series(nsw)
        Rg = eng.positions()[pick].copy()
    assert np.all(rel(E0, Erec, 1.0) < TOL.SINGLE * 10)          # the same states (total_energy of either handle)
    s = sys_of(O, p2)
    with ThreadPoolExecutor(min(16, len(os.sched_getaffinity(0)))) as ex:
        refs = list(ex.map(lambda r: O.chain(s, int(seeds[r]), Rall[r], O.W_FIXTURE, T, A, 0, nsw, 1), pick))
    for r, ref in zip(pick, refs):
        assert np.array_equal(jj[r], ref["jj"]) and ref["jj"].min() > 1000, (r, jj[r], ref["jj"])
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        assert np.array_equal(ob2["zhist"][r], ref["zhist"]), r
    TOL.assert_positions(Rg, np.stack([ref["R"] for ref in refs]), nsw, "full occupancy from per-replica states")


@pytest.mark.parametrize("N,lat,extra,kernel", [
    (4096, (8, 16), 37, "mc64"),              # the headline kernel: 4096 resident replicas + 37
    (1000, (5, 10), 300, "mc16"),             # 16 cells per lane: 6144 resident
    (2048, (8, 8), 111, "mc32"),              # 32 cells per lane: 5120 resident
    (16384, (16, 16), 5, "mt64x8"),           # two teams of wavefronts: 256 resident (config 5's kernel: 261 replicas)
    (6144, (16, 6), 20, "mc32x4"),            # four wavefronts per replica
])
def test_replica_counts_that_do_not_fill_the_device_run_as_windows_of_units(S, O, N, lat, extra, kernel):
    """Round 5 (VERDICT r4 "what's weak" #12): with nrep = G + r replicas (G = what the device holds at once,
    smcx_replica_granule) a launch of all replicas costs two rounds, the second nearly empty.  The sweeps between two gathers
    now run as WINDOWS of G (replica, block) units (csrc/smcx_sweep_ma.hip: MaArgs2) -- the replicas of one launch sit in
    different blocks of the group.  Replicas are independent chains (SMC.c:40, 66-95), so nothing may change: (a) every replica
    below G gives BIT-IDENTICAL results to the same replica in a handle of exactly G replicas (plain launches); (b) replicas
    around the window boundaries and the last ones equal the oracle chain.  4 sweeps with a gather before the last: a group of
    three sweeps (blocks of 2 + 1: a short tail block, launches that span two blocks) and a group of one."""
    R0 = O.fcc(*lat).reshape(-1, 3)[:N].ravel().copy()
    nsw, gl = 4, 4
    geom = {"mt64x8": (64, 8)}.get(kernel, (0, 0))
    # (a probe handle for G; more than 1024 replicas of N <= 1024, where fewer would get sweep_kernel_ml16 and its own G)
    p1 = S.default_params(N, 1100 if N <= 1024 else 8, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=geom[0], tune_waves=geom[1])
    with S.Engine(p1) as eng:
        G, _ = eng.replica_granule()
    assert G > 0
    nrep = G + extra
    out = {}
    for tag, n in (("windows", nrep), ("plain", G)):
        p = S.default_params(N, n, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=geom[0], tune_waves=geom[1])
        with S.Engine(p) as eng:
            assert eng.kernel_form[1] == "smcx::sweep_kernel_" + kernel, eng.kernel_form
            g2, note = eng.replica_granule()
            assert g2 == G and (("windows of" in note) == (tag == "windows")), note
            eng.upload(R0, O.W_FIXTURE)
            eng.run(0, nsw, gl)
            E, jj = eng.series(nsw)
            ob = eng.observables()
            out[tag] = dict(E=E, jj=jj, zh=ob["zhist"], acc=ob["accepted"], R=eng.positions())
    a, b = out["windows"], out["plain"]
    for k in ("E", "jj", "zh", "acc", "R"):
        assert np.array_equal(a[k][:G], b[k]), k                 # (a) bit-identical, every replica of the full window
    pick = sorted(set([0, G - 1, G, G + 1, nrep - 2, nrep - 1]))  # (b) the oracle at the seams and in the remainder
    s = sys_of(O, p)
    with ThreadPoolExecutor(min(len(pick), len(os.sched_getaffinity(0)))) as ex:
        refs = list(ex.map(lambda r: O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, nsw, gl), pick))
    for r, ref in zip(pick, refs):
        assert np.array_equal(a["jj"][r], ref["jj"]), (r, a["jj"][r], ref["jj"])
        TOL.assert_series(a["E"][r], ref["E"], what="replica %d" % r)
        assert np.array_equal(a["zh"][r], ref["zhist"]), r
    assert a["jj"].sum() > 0


@pytest.mark.parametrize("N,lat,slots,waves,kernel", [(4096, (8, 16), 0, 0, "mc64"), (4096, (8, 16), 64, 1, "mb64"),
                                                      (16384, (16, 16), 64, 8, "mt64x8"), (16384, (16, 16), 64, 4, "mc64x4"),
                                                      (16384, (16, 16), 32, 8, "mc32x8")])
def test_last_particle_in_the_top_corner_cell(S, O, N, lat, slots, waves, kernel):
    """Regression (round 5).  zsort_kernel marked empty cells by the all-ones key and tested "key != ~0u" for "holds a particle";
    but the SECOND sort key -- (group | Morton code of x, y | particle index) -- of the last particle (N - 1 with N = 4096 or
    16384: all index bits set) in the top z group with the highest Morton code (x, y in the (+L/2, +L/2) corner cell of the
    256 x 256 grid) IS all ones: that particle was taken for an empty cell, its entry of the cell-ordered positions zeroed
    (a phantom particle at the origin for every probe within the cutoff of (0, 0, 0); the real particle invisible to the
    others) until the next sort.  Short runs from the lattice never met it (the fcc start puts the last particle ~1 away from
    that corner; it diffuses in after ~250 sweeps: profiles/r05_zsort_sentinel_bug.txt -- found through the incremental-energy
    check of the 500-sweep test above).  Here the last particle is PUT there: two sweeps against the oracle (SMC.c:278-351),
    accepted counts equal, energies within the schedule; with the round-4 kernel the energy of sweep 1 is off by O(1)."""
    R0 = O.fcc(*lat).reshape(-1, 3).copy()
    top = R0[:, 2].max()
    R0[N - 1] = [16.47, 16.47, top + 1.0]          # the corner cell (x, y > L/2 - L/256), above everything else
    assert np.sum(np.linalg.norm(R0[:N - 1], axis=1) < 3.0) > 0, "a particle within the cutoff of the origin must exist"
    R0 = R0.ravel()
    nrep, nsw = 2, 2
    kw = {"tune_kernel": S.KERNEL_MB} if kernel == "mb64" else {}
    p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=slots, tune_waves=waves, **kw)
    with S.Engine(p) as eng:
        assert eng.kernel_form[1] == "smcx::sweep_kernel_" + kernel, eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 1)
        Es, jj = eng.series(nsw)
        ob = eng.observables()
        Erec = eng.total_energy()
    refs = oracle_chains(O, sys_of(O, p), [12345 + r for r in range(nrep)], R0, 0, nsw, 1)
    for r, ref in enumerate(refs):
        assert np.array_equal(jj[r], ref["jj"]), (r, jj[r], ref["jj"])
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
    assert np.all(rel(ob["E_last"], Erec, 1.0) < TOL.INCREMENTAL)


def test_incremental_energy_equals_recomputed_energy_over_long_runs_of_every_kernel_family():
    """tools/soak_energy.py --quick: twelve cases (mc64 into the walls, ml16, mc16 ragged, mc32, mb64, mc32x4, mt64x8, mc64x4,
    mc32x8, a dense gas in a small box, no walls, a replica count that runs as windows of units), 60-800 sweeps each in chunks:
    after every chunk the energy carried along the chain (SMC.c:340-341) equals the energy recomputed from the positions
    (SMC.c:626-646, 822-859) to 1e-9 relative for EVERY replica.  The invariant that found the z sort's bug in round 5; the full
    length (3000 sweeps at 4096 x 4096, 4.2e7 replica-sweeps in all: worst 6e-13) is profiles/r05_soak_energy.txt."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_energy.py"), "--quick"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-1500:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 13 and rows[-1]["ok"], rows[-1]
    for row in rows[:-1]:
        assert row["kernel"] == row["expected_kernel"], row
        assert row["count_beyond_1e-9"] == 0 and row["max_relative_incremental_minus_recomputed"] < TOL.INCREMENTAL, row


# ------------------------------------------------------------------ BASELINE config 2: N = 1024 x 1024
def test_config2_N1024_x1024(S, O):
    """BASELINE configs[1] at its real replica count: N=1024 + wall, 1024 replicas on one GPU.
    Invariants over all replicas, and eight replicas spread over the ensemble against the oracle."""
    R0 = O.fcc(8, 4)
    nrep, nsw = 1024, 5
    p = S.default_params(1024, nrep, flags=S.FLAG_WALLS | S.FLAG_SERIES)
    with S.Engine(p) as eng:
        form, name = eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 1)
        ob = eng.observables()
        Es, jj = eng.series(nsw)
        g, oob = eng.hist_info()
        Erec = eng.total_energy()
    # (round 3: two teams of one wavefront each, mt16x2; round 4: one wavefront, both probes in one pass, positions in LDS)
    assert form == 2 and name == "smcx::sweep_kernel_ml16", name
    assert np.all(rel(ob["E_last"], Erec, 1.0) < TOL.INCREMENTAL)
    assert np.all(g == nsw) and np.all(oob == 0) and np.all(ob["zhist"].sum(axis=1) == nsw * 1024)
    pick = sorted(set([0, 1, 1022, 1023] + [int(x) for x in np.linspace(0, 1023, 126).round()]))   # 128 replicas: ~8 s of oracle
    refs = oracle_chains(O, sys_of(O, p), [12345 + r for r in pick], R0, 0, nsw, 1)
    for r, ref in zip(pick, refs):
        assert np.array_equal(jj[r], ref["jj"]), r
        TOL.assert_series(Es[r], ref["E"], what="replica %d" % r)
        assert np.array_equal(ob["zhist"][r], ref["zhist"]), r


# ------------------------------------------------------------------ the screen never drops a pair (device)
def _ab_kernels(S, O, N, lat, nrep, nsw, mx_geom, fp_geom):
    R0 = O.fcc(*lat)
    out = []
    for kernel, (slots, waves) in ((1, fp_geom), (2, mx_geom)):
        p = S.default_params(N, nrep, tune_kernel=kernel, tune_slots=slots, tune_waves=waves,
                             flags=S.FLAG_WALLS | S.FLAG_SERIES)
        with S.Engine(p) as eng:
            assert eng.kernel_form[0] == kernel
            eng.upload(R0, O.W_FIXTURE)
            eng.run(0, nsw, nsw)
            E, jj = eng.series(nsw)
            out.append((E.copy(), jj.copy()))
    return out


@pytest.mark.parametrize("N,lat,nrep,nsw,mx_geom,fp_geom", [
    (4096, (8, 16), 256, 10, (64, 1), (16, 4)),     # the benchmark kernel, 1.0e7 moves
    (16384, (16, 16), 32, 2, (64, 4), (32, 8)),     # config 5, dense: ~60 pairs inside the cutoff per probe
])
def test_screen_ab_against_fp64_kernel_at_scale(S, O, N, lat, nrep, nsw, mx_geom, fp_geom):
    """screened kernel vs all-fp64 kernel from the same start and seeds.  The two sum in a different
    order, so they agree to rounding until chaos amplifies it: measured on MI355X at N=4096 the energy
    difference grows about tenfold per sweep from 1e-14 relative (2e-13 after five sweeps, 2e-8 after
    ten) and eventually flips an accept decision.  A pair dropped by the screen would instead shift E
    by >= 4|V(rc)| = 5e-3 at once.  Required: through sweep four |dE| within the schedule of tests/tolerances.py for
    every replica (4e-9 .. 1e-3 + 1e-11 |E|: the z-ordered cells of sweep_kernel_mb64 / mc64 sum in yet another order: 3e-6 at
    most after four sweeps, measured over 256 replicas) and equal accept counts over those sweeps; over all sweeps at most
    2 % of the replicas with a differing accept count.  (The counters of the diagnostic build, below, test
    every cell on every move directly.)"""
    (Ea, ja), (Eb, jb) = _ab_kernels(S, O, N, lat, nrep, nsw, mx_geom, fp_geom)
    k = min(nsw, 4)
    TOL.assert_series(Ea[:, :k + 1], Eb[:, :k + 1], what="screened against fp64 kernel")
    assert np.array_equal(ja[:, :k], jb[:, :k]) and ja.sum() > 0
    diverged = int((ja != jb).any(axis=1).sum())
    assert diverged <= max(1, nrep // 50), diverged


def _run_form(S, N, lat, nrep, nsw, slots, kernel=0, resort=0):
    """one run of `nsw` sweeps through the sweep-kernel form `kernel` (smcx_params.tune_kernel; the plan is per
    handle, so the forms are compared inside one process)"""
    p = S.default_params(N, nrep, flags=S.FLAG_WALLS | S.FLAG_SERIES, tune_slots=slots, tune_waves=1,
                         tune_kernel=kernel, tune_resort=resort)
    with S.Engine(p) as eng:
        name = eng.kernel_form[1]
        eng.upload(S.fcc_init(*lat), S.W_REFERENCE)
        eng.run(0, nsw, nsw)
        E, jj = eng.series(nsw)
        return dict(E=E, jj=jj, R=eng.positions(), name=name)


@pytest.mark.parametrize("N,lat,nrep,nsw,slots", [(4096, (8, 16), 4096, 3, 64), (4000, (10, 10), 64, 3, 64),
                                                  (1024, (8, 4), 1024, 4, 16), (2048, (8, 8), 256, 3, 32),
                                                  (1960, (7, 10), 64, 3, 32), (720, (6, 5), 64, 4, 16)])
def test_hand_scheduled_kernel_matches_compiled_kernel(S, O, tmp_path, N, lat, nrep, nsw, slots):
    """sweep_kernel_ma (hand-scheduled, the benchmark's kernel) against sweep_kernel_mi (the same algorithm
    compiled by hipcc, whose screen the miss counter of the diagnostic build validates) on the whole bench
    workload: 4096 replicas x 3 sweeps = 5e7 moves, a ragged N with padding slots, and the 16- and 32-particles-per-lane variants that serve
    N <= 2048 (BASELINE configs[1] is N=1024).  Both evaluate the
    same pairs in the same lanes and rounds; they differ in the Metropolis arithmetic's association
    (row layout), i.e. by rounding.  A pair missed by either screen would shift E by >= 5e-3."""
    legs = [("ma", S.KERNEL_MA), ("mi", S.KERNEL_MI), ("mc", S.KERNEL_AUTO)] + ([("mb", S.KERNEL_MB)] if slots == 64 else [])
    out = {tag: _run_form(S, N, lat, nrep, nsw, slots, kernel) for tag, kernel in legs}
    assert str(out["ma"]["name"]) == "smcx::sweep_kernel_ma%d" % slots and "sweep_kernel_mi" in str(out["mi"]["name"])
    # the z-ordered forms (cells in z order, only the groups in reach screened): the default for this box, one word
    # per cell (int8 x, y + int16 z) screened by v_dot4_i32_i8; and with 64 particles per lane also int16 x,y in
    # registers + int16 z in LDS
    # (few replicas of N <= 1024: the form of mc16 with the cells' positions in LDS)
    assert str(out["mc"]["name"]) == ("smcx::sweep_kernel_ml16" if slots == 16 and nrep <= 1024 else "smcx::sweep_kernel_mc%d" % slots)
    if slots == 64:
        assert str(out["mb"]["name"]) == "smcx::sweep_kernel_mb64"
    for tag in out:
        if tag == "mi":
            continue
        assert np.array_equal(out[tag]["jj"], out["mi"]["jj"]) and out[tag]["jj"].sum() > 0, tag
        # energies of every sweep and final positions (worst and median replica): the schedule of tests/tolerances.py, which
        # was derived from exactly this comparison (profiles/r04_rounding_drift_two_kernels.txt)
        TOL.assert_series(out[tag]["E"], out["mi"]["E"], what=tag)
        TOL.assert_positions(out[tag]["R"], out["mi"]["R"], nsw, tag)


def test_several_sweeps_per_launch_between_sorts(S, O, tmp_path):
    """tune_resort = 3: the z sort runs every third sweep and the sweep kernel loops over the sweeps of a launch itself
    (group ranges widened by three sweeps of moves, energy carried in a register, per-sweep records): same chains
    as with a sort before every sweep, to rounding (the cells differ, so the sums associate differently)."""
    out = {tag: _run_form(S, 4096, (8, 16), 64, 7, 64, resort=resort) for tag, resort in (("every", 1), ("third", 3))}
    assert str(out["third"]["name"]) == "smcx::sweep_kernel_mc64"
    k = 4
    assert np.array_equal(out["every"]["jj"][:, :k], out["third"]["jj"][:, :k]) and out["every"]["jj"].sum() > 0
    TOL.assert_series(out["every"]["E"][:, :k + 1], out["third"]["E"][:, :k + 1], what="sort every third sweep")
    assert np.abs(out["every"]["E"] - out["third"]["E"]).max() < TOL.CAP_ENERGY


def test_benchmark_kernel_ensemble_statistics_beyond_chaos_horizon(S, O, tmp_path):
    """sweep_kernel_mc64 against sweep_kernel_mi (hipcc-compiled, every slot screened) on the benchmark's system, 256
    replicas x 60 sweeps: the chains separate after ~10 sweeps (chaos), so the realised energies differ; the ensemble
    means of the final energy and of the accepted moves must agree within 4 standard errors of their difference.
    A bias from a rare path of the z-ordered kernel (group ranges, lane assignment, issue priorities are all
    configuration dependent) would accumulate here."""
    out = {tag: _run_form(S, 4096, (8, 16), 256, 60, 64, kernel) for tag, kernel in (("mc", S.KERNEL_AUTO), ("mi", S.KERNEL_MI))}
    assert str(out["mc"]["name"]) == "smcx::sweep_kernel_mc64" and "sweep_kernel_mi" in str(out["mi"]["name"])
    for what, a, b in (("final energy", out["mc"]["E"][:, -1], out["mi"]["E"][:, -1]),
                       ("accepted moves", out["mc"]["jj"].sum(axis=1).astype(float), out["mi"]["jj"].sum(axis=1).astype(float))):
        se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        print("%s: %.4f vs %.4f (difference %.2f standard errors)" % (what, a.mean(), b.mean(), (a.mean() - b.mean()) / se))
        assert abs(a.mean() - b.mean()) < 4 * se, what
    # and they did separate: identical trajectories would make the comparison vacuous
    assert (out["mc"]["jj"] != out["mi"]["jj"]).any()


def _load_check_build():
    # (SMCX_CHECK_LIB: the diagnostic build of a VARIANT library, for A/B sessions)
    path = os.environ.get("SMCX_CHECK_LIB") or os.path.join(ROOT, "montecarlo-surfacer_amd", "libsmcx_check.so")
    if not os.path.exists(path):
        pytest.fail("libsmcx_check.so is not built (make -C montecarlo-surfacer_amd/csrc CHECK=1; build() does it)")
    old = os.environ.get("SMCX_LIB")
    os.environ["SMCX_LIB"] = path
    try:
        spec = importlib.util.spec_from_file_location(
            "montecarlo_surfacer_amd_check", os.path.join(ROOT, "montecarlo-surfacer_amd", "__init__.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod._lib()
    finally:
        if old is None:
            del os.environ["SMCX_LIB"]
        else:
            os.environ["SMCX_LIB"] = old
    return mod


@pytest.mark.parametrize("N,lat,nrep,nsw,slots,waves", [
    (4096, (8, 16), 4096, 2, 0, 0),      # THE bench workload: 4096 replicas, z as fp16 in LDS (3.4e7 moves)
    (16384, (16, 16), 64, 1, 64, 4),     # config 5, fp32 z, dense
    (1024, (8, 4), 1024, 4, 0, 0),       # config 2
])
def test_screen_miss_counter_is_zero(O, N, lat, nrep, nsw, slots, waves):
    """diagnostic build (libsmcx_check.so = the product's sources with -DSMCX_CHECK): beside the screen
    the kernel runs the fp64 cutoff test on EVERY slot and probe from the fp64 positions in memory and
    counts the pairs inside the cutoff that the screen did not flag.  Must be zero; the counts of true
    hits and of candidates show the check is not vacuous and how tight the screen is."""
    import ctypes as C
    K = _load_check_build()
    p = K.default_params(N, nrep, tune_slots=slots, tune_waves=waves)
    with K.Engine(p) as eng:
        assert eng.kernel_form[0] == 2
        eng.upload(K.fcc_init(*lat), K.W_REFERENCE)
        eng.run(0, nsw, nsw)
        cnt = (C.c_uint64 * 3)()
        f = K._lib().smcx_debug_check_counts
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert f(eng._h, cnt) == 0
        acc = eng.observables()["accepted"].sum()
    inside, cand, miss = int(cnt[0]), int(cnt[1]), int(cnt[2])
    moves = nrep * nsw * N
    print("N=%d: %d moves, %d pairs inside the cutoff, %d candidates (%.2fx), %d missed" %
          (N, moves, inside, cand, cand / max(inside, 1), miss))
    assert miss == 0
    assert inside > moves and cand >= inside and acc > 0
    assert cand < 3 * inside + 40 * moves      # and the screen still screens


_MBC_WORKER = r"""
import sys, os, ctypes as C, importlib.util, json
root = sys.argv[1]
os.environ["SMCX_LIB"] = os.path.join(root, "montecarlo-surfacer_amd", "libsmcx_check.so")
spec = importlib.util.spec_from_file_location("smcx_chk", os.path.join(root, "montecarlo-surfacer_amd", "__init__.py"))
K = importlib.util.module_from_spec(spec); spec.loader.exec_module(K)
N, Na, Nz, nrep, nsw, gl, slots, waves = (int(v) for v in sys.argv[2:10])
L, A = (float(v) for v in sys.argv[10:12]) if len(sys.argv) > 11 else (33.0, 1.1)     # box width, step size (condensed states)
p = K.default_params(N, nrep, tune_slots=slots, tune_waves=waves, L=L, A=A)
with K.Engine(p) as eng:
    name = eng.kernel_form[1]
    eng.upload(K.fcc_init(Na, Nz, L=L), K.W_REFERENCE)
    eng.run(0, nsw, gl)
    cnt = (C.c_uint64 * 8)()
    f = K._lib().smcx_debug_work_counts
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert f(eng._h, cnt) == 0
    acc = int(eng.observables()["accepted"].sum())
print(json.dumps({"name": name, "inside": int(cnt[0]), "cand": int(cnt[1]), "miss": int(cnt[2]), "groups": int(cnt[3]),
                  "passes": int(cnt[4]), "rounds": int(cnt[5]), "unworked": int(cnt[7]), "acc": acc}))
"""


def _run_check_worker(tmp_path, mode, N, lat, nrep, nsw, gl, slots=64, waves=1, L=33.0, A=1.1):
    w = tmp_path / "mbc_worker.py"
    w.write_text(_MBC_WORKER)
    r = subprocess.run([sys.executable, str(w), ROOT, str(N), str(lat[0]), str(lat[1]), str(nrep), str(nsw), str(gl),
                        str(slots), str(waves), repr(L), repr(A)],
                       env=dict(os.environ, SMCX_CHECK_MB=mode), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("N,lat,nrep,nsw,gl", [(4096, (8, 16), 256, 6, 3), (4000, (10, 10), 64, 4, 1), (2100, (5, 21), 64, 3, 3)])
def test_ranged_screen_sets_every_bit_of_the_full_screen(tmp_path, N, lat, nrep, nsw, gl):
    """sweep_kernel_mb64 screens only the 4-slot groups whose z range can reach the probe.  Its diagnostic build
    (libsmcx_check.so with SMCX_CHECK_MB=1) follows EVERY ranged pass with the full pass of sweep_kernel_ma64 for
    the same probe and counts the bits of the full result that the ranged result lacks: must be zero over all
    moves -- with re-sorts between the sweeps of a launch (gl = 3: ranges widened by two sweeps of accepted
    moves), a ragged N, and a thin tall start (fcc(5,21): few particles per z range)."""
    d = _run_check_worker(tmp_path, "1", N, lat, nrep, nsw, gl)
    moves = nrep * nsw * N
    print("N=%d: %d moves, %d candidate bits of the full passes, %d missing from the ranged passes" %
          (N, moves, d["cand"], d["miss"]))
    assert d["name"] == "smcx::sweep_kernel_mb64"
    assert d["miss"] == 0
    assert d["cand"] > moves and d["acc"] > 0


@pytest.mark.parametrize("N,lat,nrep,nsw,gl,slots,waves,kernel", [
    (4096, (8, 16), 64, 4, 2, 64, 1, "mc64"), (4000, (10, 10), 32, 3, 1, 64, 1, "mc64"), (2100, (5, 21), 32, 3, 3, 64, 1, "mc64"),
    (4096, (16, 4), 16, 2, 1, 64, 1, "mc64"),
    (1024, (8, 4), 64, 4, 2, 16, 1, "ml16"), (1000, (5, 10), 1100, 2, 1, 16, 1, "mc16"),        # N <= 1024: few replicas (positions
                                                                                                # in LDS) / many
    (2048, (8, 8), 32, 3, 1, 32, 1, "mc32"),
    (16384, (16, 16), 4, 2, 1, 64, 4, "mc64x4"), (16384, (16, 16), 4, 2, 1, 32, 8, "mc32x8"),   # config 5's kernels
    (9000, (15, 10), 4, 2, 2, 64, 4, "mc64x4"), (10000, (10, 25), 4, 2, 1, 32, 8, "mc32x8"),    # ragged, tall
    (8192, (16, 8), 4, 2, 1, 32, 4, "mc32x4"), (4800, (10, 12), 4, 2, 2, 32, 4, "mc32x4"),      # 4096 < N <= 8192
    (16384, (16, 16), 4, 2, 1, 64, 8, "mt64x8"), (9000, (15, 10), 4, 2, 1, 64, 8, "mt64x8")])    # two teams of wavefronts
def test_byte_screen_kernel_misses_no_pair_inside_the_cutoff(tmp_path, N, lat, nrep, nsw, gl, slots, waves, kernel):
    """sweep_kernel_mc64 (the benchmark's kernel), mc16 / mc32 (N <= 2048) and the several-wavefront forms mc64x4 /
    mc32x8 (8192 < N <= 16384): one word per cell screened by v_dot4_i32_i8, only the groups in
    z reach.  The diagnostic build (SMCX_CHECK_MB=2) runs, beside EVERY pass, the fp64 cutoff test with the
    minimum image of SMC.c:567-578 on EVERY cell from the fp64 positions in memory and counts the pairs inside
    the cutoff whose bit the pass did not set (the moving particle and the probe's own particle excepted, as
    in the reference's loop): must be zero; the counts of true pairs and of candidate bits show the check is not
    vacuous and how tight the screen is (the byte units flag about 1.5x the cutoff sphere).  Last case: the dense
    film fcc(16,4), ~60 pairs inside the cutoff per probe."""
    d = _run_check_worker(tmp_path, "2", N, lat, nrep, nsw, gl, slots, waves)
    moves = nrep * nsw * N
    print("N=%d %s: %d moves, %d pairs inside the cutoff, %d candidate bits (%.2fx), %d missed; %.2f groups of %d per pass, "
          "%.2f passes per move and wavefront" %
          (N, kernel, moves, d["inside"], d["cand"], d["cand"] / max(d["inside"], 1), d["miss"],
           d["groups"] / max(d["passes"], 1), slots // 4, d["passes"] / (moves * waves)))
    assert d["name"] == "smcx::sweep_kernel_" + kernel
    assert d["miss"] == 0
    # the executed-work counters: two passes per move and wavefront (+ one per run start), at most all groups each
    # (the two-team kernels: ONE pass per move and wavefront)
    ppm = 1 if kernel.startswith("mt") else 2
    assert ppm * moves * waves <= d["passes"] <= ppm * (moves + 4 * nrep * nsw) * waves
    assert 0 < d["groups"] <= d["passes"] * (slots // 4)
    assert d["inside"] > moves and d["cand"] >= d["inside"] and d["acc"] > 0
    assert d["cand"] < 3 * d["inside"] + 40 * moves


@pytest.mark.parametrize("L,A", [(25.6, 0.004), (18.0, 4e-5)])
def test_two_team_list_hands_every_item_to_a_working_lane(tmp_path, L, A):
    """sweep_kernel_mt64x8 in a CONDENSED state (fcc(16,16) with a = 1.6: the Lennard-Jones crystal, ~160 candidate bits per
    probe) and an OVERFULL one (a = 1.125: ~100 bits per wavefront and probe, both hand-overs fill the list).  Round 4's list of
    64 entries let s_bfm_b64's 6-bit count wrap to an EMPTY mask of working lanes when exactly 64 items were handed over from
    lane 0 (every wave without special lanes): all 64 candidates dropped, their bits already out of the flag words, nothing
    noticed.  The diagnostic build now counts, at every hand-over, the items for which no working lane is enabled
    (`unworked`, must be 0; with the round-4 generator, SMCX_GEN_TTCAP=64, this case counts thousands:
    profiles/r05_two_team_list_overflow.txt) beside the screen's own miss count; `rounds` shows that the full-list path ran."""
    d = _run_check_worker(tmp_path, "2", 16384, (16, 16), 2, 2, 1, 64, 8, L=L, A=A)
    print("L=%.1f: %d pairs inside the cutoff, %d candidate bits, %d missed by the screen, %d further rounds, %d items without a "
          "working lane, %d accepted" % (L, d["inside"], d["cand"], d["miss"], d["rounds"], d["unworked"], d["acc"]))
    assert d["name"] == "smcx::sweep_kernel_mt64x8"
    assert d["miss"] == 0 and d["unworked"] == 0
    assert d["acc"] > 0 and d["cand"] > 100 * 2 * 2 * 16384
    if L < 20:
        assert d["rounds"] > 2 * 16384          # the list was full on most probes: the 64th item went to a further round


def test_wavefront_lifetimes_of_a_launch_are_reported(S):
    """bench workload, sweep_kernel_mc64: 4096 wavefronts start together, four per SIMD.  Without the priority
    table the arbiter serves the oldest wavefront of a SIMD first and the lifetimes of one launch
    span 7.3 .. 13.2 ms (the launch lasts as long as the slowest); with it they were within 8 % of the median on
    the round-2 box.  That spread is a performance figure (clock- and device-dependent): it is PRINTED here, and
    the parity suite asserts only that the in-kernel clock stamps are sane (every wavefront stamped, ordered)."""
    p = S.default_params(4096, 4096)
    with S.Engine(p) as eng:
        assert eng.kernel_form[1] == "smcx::sweep_kernel_mc64"
        eng.upload(S.fcc_init(8, 16), S.W_REFERENCE)
        eng.run(0, 2, 10)
        lo, med, hi, span = eng.wave_spread()
    print("wavefront lifetimes of the last launch: %.0f .. %.0f us (median %.0f), launch span %.0f" % (lo, hi, med, span))
    assert 0 < lo <= med <= hi <= span * 1.0001


# ------------------------------------------------------------------ statistics beyond the chaos horizon
@pytest.mark.parametrize("N,lat", [(256, (4, 4)), (1024, (8, 4))])
def test_ensemble_statistics_beyond_chaos_horizon(S, O, N, lat):
    """256 replicas, 20 thermalisation + 100 production sweeps, GPU vs oracle on the
    same seeds.  Trajectories separate after 20-60 sweeps (chaos), so realised chains differ; the
    ensemble mean energy, acceptance ratio and wall-normal profile must agree within 4 standard
    errors of the difference (from the replica spread of both sides).  A bias from the log-space
    acceptance test (SMC.c:329-335) or the reciprocal's last ulp would show here."""
    nrep, eq, nsw, gl = 256, 20, 100, 10
    R0 = O.fcc(*lat)
    p = S.default_params(N, nrep)
    with S.Engine(p) as eng:
        eng.upload(R0, O.W_FIXTURE)
        eng.run(eq, nsw, gl)
        ob = eng.observables()
        g, _ = eng.hist_info()
    refs = oracle_chains(O, sys_of(O, p), [12345 + r for r in range(nrep)], R0, eq, nsw, gl)

    def check(name, a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        assert abs(a.mean() - b.mean()) <= 4 * se + TOL.RATIO * abs(b.mean()), (name, a.mean(), b.mean(), se)
        return abs(a.mean() - b.mean()) / (se + 1e-300)

    zs = [check("meanE", ob["meanE"], [r["meanE"] for r in refs]),
          check("acceptance", ob["acceptance_ratio"], [r["acceptance_ratio"] for r in refs])]
    prof_g = ob["zhist"] / g[:, None].astype(float)
    prof_c = np.array([r["zhist"] / float(r["gathers"]) for r in refs])
    occupied = 0
    for k in range(p.Ncz):
        if prof_c[:, k].sum() + prof_g[:, k].sum() > 0:
            zs.append(check("z bin %d" % k, prof_g[:, k], prof_c[:, k]))
            occupied += 1
    assert occupied >= 2 and np.all(g == nsw // gl)
    # the chains really are beyond the horizon: most replicas ended on a different trajectory
    differ = sum(int(ob["accepted"][r]) != refs[r]["accepted"] for r in range(nrep))
    print("N=%d: %d of %d replicas on a different trajectory; max deviation %.2f standard errors" %
          (N, differ, nrep, max(zs)))
    assert differ > nrep // 4


# ------------------------------------------------------------------ host side, as a process
def test_smcx_main_as_child_process(O):
    """host/smcx_main.c (the main.c:7-176 counterpart) run as its own process:
    `smcx_main eqsteps maxsteps numdata T N nrep Na Nz` -- its printed ensemble energy, acceptance
    ratio and z profile against the oracle chains of the same seeds"""
    exe = os.path.join(ROOT, "montecarlo-surfacer_amd", "smcx_main")
    r = subprocess.run([exe, "1", "8", "4", "1.1", "256", "3", "4", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout

    def after(key):
        return out.split(key, 1)[1].split("\n", 1)[0]
    E = float(after("Mean energy: ").split()[0])
    acc = float(after("Average acceptance ratio: ").split()[0])
    therm = float(after("(thermalisation ").split(")")[0])
    prof = np.array([float(v) for v in after("z profile (particles per cell per gather):").split()])
    s = O.make_sys(256)
    R0 = O.fcc(4, 4)
    refs = [O.chain(s, 12345 + k, R0, O.W_FIXTURE, 1.1, 1.1, 1, 8, 2) for k in range(3)]
    assert abs(E - np.mean([q["meanE"] for q in refs])) < TOL.OBSERVABLE * abs(E) + 2 * TOL.printed(6)      # printed with %f
    assert abs(acc - np.mean([q["acceptance_ratio"] for q in refs])) < 2 * TOL.printed(6)
    assert abs(therm - np.mean([q["therm_acceptance"] for q in refs])) < 2 * TOL.printed(6)
    ref_prof = np.sum([q["zhist"] for q in refs], axis=0) / float(sum(q["gathers"] for q in refs))
    assert prof.shape == (33,) and np.abs(prof - ref_prof).max() < 2 * TOL.printed(3)   # %0.3f
    # argument errors are reported, not crashed on
    bad = subprocess.run([exe, "1", "8"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "usage" in bad.stderr
    bad = subprocess.run([exe, "0", "4", "2", "1.1", "1024"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "Can't make the reference's crystal" in bad.stderr  # SMC.c:428


# ------------------------------------------------------------------ multi-rank launch path of bench.py
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself.  Rehearsed on the one
    GPU of this box: both ranks on device 0 (SMCX_FORCE_DEVICE), gloo instead of RCCL for the gather."""
    env = dict(os.environ, SMCX_DIST_BACKEND="gloo", SMCX_FORCE_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--N", "1024",
                        "--replicas", "64", "--steps", "3", "--warmup", "1", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1                              # rank 0 alone prints
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["observables"]["replicas_gathered"] == 128 and out["config"]["replicas_total"] == 128
    assert out["value"] > 0 and out["roofline"]["bound"] == "valu_issue" and out["roofline"]["clock_ghz"] > 1.0
    # multi-rank observability: every rank's own step time (the headline is the MAX), device times and gather time
    pr = out["per_rank"]
    for k in ("ms_per_step", "sweep_kernel_ms_per_step", "device_ms_per_step", "gather_ms"):
        assert len(pr[k]["ranks"]) == 2 and pr[k]["min"] <= pr[k]["median"] <= pr[k]["max"] and pr[k]["min"] > 0, k
    assert abs(pr["ms_per_step"]["max"] - out["ms_per_step"]) < TOL.OBSERVABLE * out["ms_per_step"]
    assert out["gather_ms"] > 0
