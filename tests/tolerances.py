"""The tolerances of the GPU parity tests, stated ONCE (tests/test_gpu_parity.py, test_gpu_configs.py, test_gpu_rare_paths.py use
nothing else; round 5, VERDICT r4 "what's weak" #1).

What is compared, and why the bound is what it is:

1. INTEGERS -- accepted moves of every sweep, histogram bins, visiting order, cluster counts: always EQUAL.  They are what
   carries parity furthest: one pair dropped from a sum shifts an energy by >= 4|V(rc)| = ONE_MISSED_PAIR and flips a
   Metropolis decision (SMC.c:326-335) within a sweep or two.

2. SINGLE EVALUATIONS (K1-K5, teacher-forced moves: no chain in between): SINGLE, relative to the value (+ a stated scale).
   The GPU sums the neighbour loop as a 64-lane tree instead of l = 0..N-1, contracts a*b+c into FMA, multiplies by 1/L and
   1/T where the reference divides and uses the device libm: ~1e-16 relative per operation.

3. FREE-RUNNING CHAINS.  Those rounding differences are amplified by the dynamics (chaos).  Measured on the
   benchmark class of systems (N = 4000 ragged, 64 replicas, two correct kernels against each other AND each against the
   oracle: tools/probes/ragged_divergence.py, profiles/r04_rounding_drift_two_kernels.txt) the largest position difference
   after sweep 1 / 2 / 3 is

        worst of 64 replicas   2.2e-11   3.1e-10   1.5e-7         (x 15 .. x 480 per sweep)
        median replica         1.1e-13   1.3e-12   1.1e-11        (x 10 per sweep)
        max |dE| (E ~ -150)    4.1e-11   3.6e-10   1.9e-7         (= 1.2 .. 1.9 x the position difference)

   so the bound is a SCHEDULE in the sweep index k (sweeps completed since the common start, thermalisation included):

        position(k)        = 1e-11 * 100^k      1e-9, 1e-7, 1e-5      worst replica; margin over the measured 45x, 320x, 67x
        position_median(k) = 1e-12 * 10^k       1e-11, 1e-10, 1e-9    median over >= 32 replicas; margin 90x, 77x, 90x
        energy(k, E)       = 4 * position(k) + 1e-11 |E|              (forces are O(1): dE ~ F dR; the second term is the
                                                                       summation-order floor of a sum of ~N^2 pair terms)

   capped at CAP_POSITION / CAP_ENERGY (ten / five times below ONE_MISSED_PAIR), reached at k = 4: from there on a pass says
   only "no pair was dropped", and the integers and the earlier sweeps of the same test carry the parity.  The schedule is
   the same for kernel-against-oracle and kernel-against-kernel comparisons (the measurement found either kernel as far
   from the oracle as from the other kernel).

4. OBSERVABLES of a chain inside the horizon (acceptance ratio, mean energy, wall-normal profile): OBSERVABLE = north_star's
   1e-6 relative.  The acceptance ratio is an integer over N * sweeps: RATIO covers the division only.

5. INCREMENTAL against recomputed energy of ONE state (SMC.c:340-341 adds Un - Um per accepted move; `total_energy` sums all
   pairs afresh): INCREMENTAL relative to 1 + |E|, for any number of sweeps -- no chaos enters, both describe the same
   positions; what accumulates is one rounding of |Un - Um| ~ 1 per accepted move (random walk: 1e-16 sqrt(moves)).
"""
import numpy as np

ONE_MISSED_PAIR = 5e-3          # 4 |V(rc)| at rc = 3: the least a dropped pair changes an energy
SINGLE = 1e-12                  # single evaluations, relative
OBSERVABLE = 1e-6               # north_star: observables within 1e-6 relative of the reference
RATIO = 1e-12                   # acceptance ratio = integer / (N sweeps)
INCREMENTAL = 1e-9              # E carried along a chain against E recomputed from the final positions, relative to 1 + |E|
EXACT_SERIES = 1e-12            # the same numbers through another path of the host code (files, gathers, C driver)
FFT = 1e-9                      # hipFFT against the restated transform of the same series (SMC.c:1055-1093), autocorrelations of O(1)

POS_FLOOR, POS_GROWTH = 1e-11, 100.0
MED_FLOOR, MED_GROWTH = 1e-12, 10.0
ENERGY_PER_POSITION = 4.0
ENERGY_FLOOR_REL = 1e-11
CAP_POSITION = ONE_MISSED_PAIR / 10
CAP_ENERGY = ONE_MISSED_PAIR / 5
HORIZON = 3                     # sweeps up to which the schedule is below the caps by >= 50x: where value-by-value parity is claimed


def position(k):
    """bound on max |R_a - R_b| of ONE replica after k sweeps from a common start"""
    return min(POS_FLOOR * POS_GROWTH ** max(int(k), 0), CAP_POSITION)


def position_median(k):
    """bound on the MEDIAN over >= 32 replicas of max |R_a - R_b| after k sweeps"""
    return min(MED_FLOOR * MED_GROWTH ** max(int(k), 0), CAP_POSITION)


def energy(k, E):
    """bound on |E_a - E_b| of the running energy after k sweeps (array-valued in k and E)"""
    k = np.maximum(np.asarray(k, dtype=float), 0.0)
    chaos = np.minimum(ENERGY_PER_POSITION * POS_FLOOR * POS_GROWTH ** k, CAP_ENERGY)
    return chaos + ENERGY_FLOOR_REL * np.abs(np.asarray(E, dtype=float))


def series_ok(Ea, Eb, k0=0):
    """energy series [.., nsw + 1] (entry j = after k0 + j sweeps) of two arms within the schedule; returns (ok, worst ratio)"""
    Ea, Eb = np.asarray(Ea, dtype=float), np.asarray(Eb, dtype=float)
    k = k0 + np.arange(Ea.shape[-1])
    ratio = np.abs(Ea - Eb) / energy(k, Eb)
    return bool(np.all(ratio <= 1.0)), float(np.max(ratio))


def assert_series(Ea, Eb, k0=0, what=""):
    ok, worst = series_ok(Ea, Eb, k0)
    assert ok, "%s: energy series outside the schedule (worst |dE| / bound = %.3g)" % (what, worst)


def assert_energy(Ea, Eb, k, what=""):
    d = np.abs(np.asarray(Ea, dtype=float) - np.asarray(Eb, dtype=float))
    b = energy(k, Eb)
    assert np.all(d <= b), "%s: |dE| %.3g > %.3g after %d sweeps" % (what, float(np.max(d)), float(np.min(b)), k)


def assert_mean_energy(ma, mb, k, what=""):
    """mean of the series E[0..k]: no worse than its last entry"""
    assert_energy(ma, mb, k, what + " (mean energy)")


def assert_positions(Ra, Rb, k, what=""):
    """positions [.., 3N] of one or many replicas after k sweeps: every replica within position(k); with >= 32 replicas also
    the median replica within position_median(k)"""
    d = np.abs(np.asarray(Ra) - np.asarray(Rb))
    d = d.reshape(-1, d.shape[-1]).max(axis=1) if d.ndim > 1 else d.max(keepdims=True)
    assert d.max() <= position(k), "%s: max |dR| %.3g > %.3g after %d sweeps" % (what, d.max(), position(k), k)
    if len(d) >= 32:
        assert np.median(d) <= position_median(k), "%s: median |dR| %.3g > %.3g after %d sweeps" % (what, np.median(d), position_median(k), k)


def virial(k, P):
    """bound on |P_a - P_b| of the virial pressure (SMC.c:696-720, 862-895) of the state after k sweeps: the sum of r.F terms
    moves with the positions like the energy does, relative to its own size"""
    k = np.maximum(np.asarray(k, dtype=float), 0.0)
    return (np.minimum(POS_FLOOR * POS_GROWTH ** k, CAP_POSITION) + ENERGY_FLOOR_REL) * np.abs(np.asarray(P, dtype=float))


def assert_virial(Pa, Pb, k, what=""):
    d = np.abs(np.asarray(Pa, dtype=float) - np.asarray(Pb, dtype=float))
    assert np.all(d <= virial(k, Pb)), "%s: |dP| %s > %s" % (what, d, virial(k, Pb))


def printed(decimals):
    """half a unit of the last digit of a value printed with that many decimals (files, console, SURVEY's probe values)"""
    return 0.5000001 * 10.0 ** -decimals


def rel(a, b, scale=0.0):
    return np.abs(np.asarray(a) - np.asarray(b)) / (np.abs(np.asarray(b)) + scale + 1e-300)
