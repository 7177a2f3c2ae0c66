"""CPU tests: the oracle against every pin available for this path.

The reference ships no tests or fixtures.  The pins in this file: the real
glibc rand(), the real reference matematicose.c (golden file generated from oracle/_ref), and
outputs of the real reference recorded in SURVEY.md (tests/golden/reference_pins.json).  The
bit-for-bit pin of every hot-path function on the real SMC.c / SMC_noMPI_noWall.c (compiled where
they lie into oracle/_ref) is tests/test_ref_pin.py.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def unhex(v):
    return np.array([float.fromhex(s) for s in v])


PINS = load("reference_pins.json")


# ---------------------------------------------------------------- R: rand()
def test_rand_matches_golden_glibc(O):
    g = load("glibc_rand.json")["first100"]
    for seed, vals in g.items():
        r = O.Rng(int(seed))
        assert list(r.draws(100)) == vals, "seed %s" % seed


def test_rand_matches_live_libc(O):
    libc = C.CDLL("libc.so.6")
    for seed in (3, 99991, 12345 + 4095):
        libc.srand(C.c_uint(seed))
        ref = [libc.rand() for _ in range(5000)]
        assert list(O.Rng(seed).draws(5000)) == ref


def test_rand_survey_pin(O):
    assert sorted(O.Rng(12345).draws(3)) == sorted(PINS["rand_12345"]["values"])


# ---------------------------------------------------------------- R: Box-Muller
def test_box_muller_matches_real_reference(O):
    for case in load("matematicose_ref.json")["vecBoxMuller"]:
        r = O.Rng(case["seed"])
        A = r.box_muller(float.fromhex(case["sigma"]), case["length"],
                         fill=float.fromhex(case["prefill"]))
        exp = unhex(case["out"])
        assert np.array_equal(A, exp), case  # bit-exact, odd tail untouched
        assert r.rand() == case["next_rand"]  # same number of rand() calls


def test_box_muller_live_against_ref_build(O):
    so = os.path.join(O.ORACLE_DIR, "_ref", "libmatematicose_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = C.CDLL(so)
    libc = C.CDLL("libc.so.6")
    dp = C.POINTER(C.c_double)
    ref.vecBoxMuller.argtypes = [C.c_double, C.c_size_t, dp]
    ref.vecBoxMuller.restype = None
    for seed, sigma, n in ((5, 1.3, 3 * 256), (12346, 2.0977, 3 * 108)):
        A = np.zeros(n)
        libc.srand(C.c_uint(seed))
        ref.vecBoxMuller(sigma, n, A.ctypes.data_as(dp))
        assert np.array_equal(O.Rng(seed).box_muller(sigma, n, fill=0.0), A)


def test_mean_variance_match_reference(O):
    st = load("matematicose_ref.json")["stats"]
    E = unhex(st["E"])
    # the oracle's chain reductions are sequential sums like matematicose.c:15-23, 50-63, 96-103
    s = 0.0
    for v in E:
        s += v
    mean = s / len(E)
    s2 = 0.0
    for v in E:
        s2 += v * v
    assert mean == float.fromhex(st["mean"])
    assert s2 / len(E) - mean * mean == float.fromhex(st["variance"])
    assert sum(st["jj"]) / len(st["jj"]) == float.fromhex(st["intmean"])


# ---------------------------------------------------------------- W, K5, S1, C pins
def test_wall_fixture(O):
    W = O.walls(M=3, uninit=0.0)
    assert np.array_equal(W, np.array(PINS["W"]["values"]))
    assert np.array_equal(W, O.W_FIXTURE)


def test_E0_pins(O):
    for c in PINS["E0"]["cases"]:
        if c["N"] > 4096:
            continue  # 16384: 1.3e8 pairs, checked in the slow test below
        X = O.fcc(c["Na"], c["Nz"])
        e = O.total_energy(O.make_sys(c["N"]), X, O.W_FIXTURE)
        assert abs(e - c["E0"]) <= 0.6 * 10 ** (int(np.floor(np.log10(abs(c["E0"])))) + 1 - PINS["E0"]["digits"]), c


def test_E0_pin_16384(O):
    c = PINS["E0"]["cases"][3]
    X = O.fcc(16, 16)
    e = O.total_energy(O.make_sys(16384), X, O.W_FIXTURE)
    assert abs(e - c["E0"]) < 1e-5


def test_chain_pin_N108_bit_exact(O):
    """20 free-running sweeps of the real reference, reproduced to the last bit."""
    pin = PINS["chain_N108"]
    X, placed = O.box_ref(108, 33.0, 200.0)
    assert placed == 108
    s = O.make_sys(108, Lz=200.0)
    assert O.lib().orc_energy(C.byref(s), X.ctypes.data_as(C.POINTER(C.c_double))) == pin["E_pp0"]
    assert O.total_energy(s, X, O.W_FIXTURE) == pin["E_wall0"]
    out = O.chain(s, 12345, X, O.W_FIXTURE, 1.1, 1.1, 0, 20, 1)
    assert out["Efinal"] == pin["E_incremental_20"]
    assert out["accepted"] == pin["accepted"]
    assert O.total_energy(s, out["R"], O.W_FIXTURE) == pin["E_recomputed_20"]
    assert out["zhist"].sum() == 108 * 20


def test_reference_lattice_rule_breaks_as_surveyed(O):
    # SMC.c:416-431 leaves particles at the origin for these N
    for N, unplaced in ((1024, 16), (4096, 96), (8192, 128)):
        X, placed = O.box_ref(N, 33.0, 240.0)
        assert N - placed == unplaced
    for N in (32, 108, 256, 500, 864, 2048, 4000, 16384):
        assert O.box_ref(N, 33.0, 240.0)[1] == N


def test_acceptance_pins(O):
    for c in PINS["acceptance"]["cases"]:
        if c["sweeps"] * c["N"] ** 2 > 1.5e9:
            continue
        X = O.fcc(c["Na"], c["Nz"])
        out = O.chain(O.make_sys(c["N"]), 12345, X, O.W_FIXTURE, 1.1, 1.1, 0, c["sweeps"], 10 ** 9)
        # The survey's timing build used -O3 -march=native (FMA): trajectories part after
        # ~60 sweeps, so long-run ratios agree statistically, not digit
        # for digit: 3 binomial sigma + the 3-digit rounding of the recorded value.
        moves = c["sweeps"] * c["N"]
        tol = 3 * np.sqrt(c["ratio"] * (1 - c["ratio"]) / moves) + 5e-4
        assert abs(out["acceptance_ratio"] - c["ratio"]) < tol, (c, out["acceptance_ratio"])


def test_nowall_pins(O):
    pin = PINS["nowall_N256"]
    N = 256
    L = np.cbrt(N / 0.1)
    X = np.zeros(3 * N)
    dp = C.POINTER(C.c_double)
    assert O.lib().orc_nw_fcc_init(N, L, X.ctypes.data_as(dp)) == N
    assert abs(O.lib().orc_nw_energy(N, X.ctypes.data_as(dp), L) - pin["E0"]) < 5e-13
    assert abs(O.lib().orc_nw_pressure(N, X.ctypes.data_as(dp), L) - pin["P0"]) < 5e-16  # :664-684
    assert abs(O.lib().orc_nw_energy_single(N, X.ctypes.data_as(dp), L, 0) - pin["energySingle0"]) < 5e-15
    assert abs(O.lib().orc_nw_energy_single(N, X.ctypes.data_as(dp), L, 1) - pin["energySingle1"]) < 5e-15
    # 500 sweeps at T=0.4, A=4e-8 (SMC_noMPI_noWall.c:80-81, 192): acceptance 0.980
    rng = O.Rng(12345)
    Rn = np.zeros(3 * N)
    j = C.c_int(0)
    for _ in range(500):
        O.lib().orc_nw_one_particle_moves(N, C.byref(rng.g), X.ctypes.data_as(dp), Rn.ctypes.data_as(dp),
                                          L, 4e-8, 0.4, C.byref(j), None)
    assert abs(j.value / (500 * N) - pin["acceptance_500"]) < 6e-4


def test_nowall_ten_sweep_trace(O):
    """BASELINE config 1 (SMC_noMPI_noWall.c path, CPU plumbing): a 10-sweep chain at N=256, rho=0.1,
    T=0.4, A=4e-8.  The reference holds no vector for it, so the trace is checked
    through what must hold: the run is deterministic; every move's bookkeeping is self-consistent
    (proposal = position + delta, accepted <=> u < ap, ap = exp(-(...)/T) from the traced Um, Un, Fm,
    Fn, SMC_noMPI_noWall.c:296-306); the energy change summed over accepted moves equals the change of
    orc_nw_energy (energy(), :573-591) between the first and the last configuration; the fixed visiting
    order 0..N-1 (:278) and the 4N draws per sweep (3N normals + N uniforms, no offset draw)."""
    N, T_, A_ = 256, 0.4, 4e-8
    L = np.cbrt(N / 0.1)
    dp = C.POINTER(C.c_double)

    def pair0(a, b):   # the pair term of energySingle (:606-616): minimum image in x, y and z, cutoff L/2
        d = a - b
        d = d - L * np.rint(d / L)
        r2 = float(d @ d)
        return 4.0 * (1.0 / r2 ** 6 - 1.0 / r2 ** 3) if r2 < L * L / 4 else 0.0

    def run():
        X = np.zeros(3 * N)
        assert O.lib().orc_nw_fcc_init(N, L, X.ctypes.data_as(dp)) == N
        rng = O.Rng(12345)
        Rn = np.zeros(3 * N)
        j = C.c_int(0)
        traces, dE = [], 0.0
        E0 = O.lib().orc_nw_energy(N, X.ctypes.data_as(dp), L)
        for _ in range(10):
            tr = np.zeros(N, dtype=O.TRACE_DTYPE)
            before = X.copy()
            O.lib().orc_nw_one_particle_moves(N, C.byref(rng.g), X.ctypes.data_as(dp), Rn.ctypes.data_as(dp),
                                              L, A_, T_, C.byref(j), tr.ctypes.data_as(C.POINTER(O.OrcMoveTrace)))
            assert np.array_equal(tr["n"], np.arange(N))
            pos = before.reshape(-1, 3)
            for m in tr:   # each move starts from the positions left by the moves before it
                n = m["n"]
                assert np.allclose(m["prop"], pos[n] + m["delta"], rtol=0, atol=1e-15)
                arg = m["Un"] - m["Um"] + 0.5 * np.dot(m["delta"], m["Fn"] + m["Fm"]) + \
                    (np.dot(m["Fn"] - m["Fm"], m["Fn"] - m["Fm"]) + 2 * np.dot(m["Fn"] - m["Fm"], m["Fm"])) * A_ / (4 * T_)
                assert abs(m["ap"] - np.exp(-arg / T_)) <= 1e-12 * max(1.0, m["ap"])
                assert bool(m["accepted"]) == (m["u"] < m["ap"])
                if m["accepted"]:
                    w = m["prop"] - L * np.rint(m["prop"] / L)   # shiftSystem wraps Rn before it is copied back
                    dE += m["Un"] - m["Um"]
                    if n != 0:   # energySingle never counts particle 0 as a neighbour (:603): add what it leaves out
                        dE += pair0(m["prop"], pos[0]) - pair0(pos[n], pos[0])
                    pos[n] = w
            assert np.allclose(pos.ravel(), X, rtol=0, atol=1e-15)
            traces.append(tr)
        E1 = O.lib().orc_nw_energy(N, X.ctypes.data_as(dp), L)
        return X, j.value, traces, E0, E1, dE, rng.rand()

    X1, j1, t1, E0, E1, dE, nxt1 = run()
    X2, j2, t2, _, _, _, nxt2 = run()
    assert np.array_equal(X1, X2) and j1 == j2 and nxt1 == nxt2
    assert all(np.array_equal(a, b) for a, b in zip(t1, t2))
    assert abs(E0 - PINS["nowall_N256"]["E0"]) < 5e-13
    assert 0.95 * 10 * N < j1 <= 10 * N
    # energySingle leaves particle 0 out of every neighbour loop (:603), energy() does not: with the (n, 0) pair
    # terms added back (pair0 above) the per-move differences sum to the change of energy()
    assert abs((E1 - E0) - dE) < 1e-9, (E1 - E0, dE)
    ref = O.Rng(12345)
    ref.draws(10 * 4 * N)
    assert nxt1 == ref.rand()


# ---------------------------------------------------------------- properties / edge cases
def test_incremental_energy_tracks_recomputed(O):
    X = O.fcc(4, 4)
    s = O.make_sys(256)
    out = O.chain(s, 7, X, O.W_FIXTURE, 1.1, 1.1, 0, 30, 1)
    assert abs(out["Efinal"] - O.total_energy(s, out["R"], O.W_FIXTURE)) < 1e-10
    assert out["zhist"].sum() == 256 * 30 and out["oob"] == 0


def test_E0_restart_quirk(O):
    """SMC.c:116-117 vs 194: production restarts the series from the pre-thermalisation energy."""
    X = O.fcc(4, 4)
    s = O.make_sys(256)
    a = O.chain(s, 11, X, O.W_FIXTURE, 1.1, 1.1, 5, 8, 2, e0_restart=True)
    b = O.chain(s, 11, X, O.W_FIXTURE, 1.1, 1.1, 5, 8, 2, e0_restart=False)
    assert a["E"][0] == a["E0"] and b["E"][0] != b["E0"]
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["jj"], b["jj"])
    # with the quirk the series is offset by exactly the thermalisation's energy change
    assert np.allclose(a["E"] - b["E"], a["E"][0] - b["E"][0], rtol=0, atol=1e-9)
    assert abs(b["Efinal"] - O.total_energy(s, b["R"], O.W_FIXTURE)) < 1e-9
    assert a["gathers"] == 4


def test_wall_clamp_and_force_consistency(O):
    s = O.make_sys(2)
    W = O.W_FIXTURE
    # beyond either wall the distance is clamped to -/+1e-4 (SMC.c:738-739)
    e_in = O.walls_energy_single(s, (0.3, -0.2, 120.0), W)
    e_out = O.walls_energy_single(s, (0.3, -0.2, 150.0), W)
    assert e_in == e_out and np.isfinite(e_in) and e_in > 1e30
    # force = -grad(energy) by central differences, away from the cutoff shell
    p = np.array([1.0, 2.0, -118.2])
    F = O.walls_force(s, p, W)
    h = 1e-6
    for c in range(3):
        d = np.zeros(3); d[c] = h
        num = -(O.walls_energy_single(s, p + d, W) - O.walls_energy_single(s, p - d, W)) / (2 * h)
        assert abs(num - F[c]) <= 1e-5 * max(1.0, abs(F[c]))


def test_pair_force_is_minus_gradient(O):
    rs = np.random.RandomState(1)
    N = 64
    s = O.make_sys(N, L=6.0, Lz=12.0)
    R = (rs.rand(N, 3) - 0.5) * np.array([6.0, 6.0, 8.0])
    R = R.ravel()
    F = O.force_single(s, R, 5)
    h = 1e-6
    for c in range(3):
        Rp, Rm = R.copy(), R.copy()
        Rp[15 + c] += h; Rm[15 + c] -= h
        num = -(O.energy_single(s, Rp, 5) - O.energy_single(s, Rm, 5)) / (2 * h)
        assert abs(num - F[c]) <= 1e-4 * max(1.0, abs(F[c]))


def test_histogram_uint8_wrap(O):
    """SMC.c:914-920: cell numbers pass through uint8_t."""
    s = O.make_sys(2)
    Nc = 33 ** 3
    D = np.zeros(Nc, dtype=np.uint64); Mu = np.zeros(Nc, dtype=np.uint64)
    Rbin = np.zeros(2, dtype=np.int32); oob = C.c_uint64(0)
    # particle 0 in cell (16,16,16); particle 1 beyond the upper wall: k = 33 -> next j row
    R = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 120.0 + 1e-9])
    O.lib().orc_local_density(C.byref(s), R.ctypes.data_as(C.POINTER(C.c_double)),
                              D.ctypes.data_as(C.POINTER(C.c_uint64)),
                              Rbin.ctypes.data_as(C.POINTER(C.c_int32)),
                              Mu.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(oob))
    v0 = 16 * 33 * 33 + 16 * 33 + 16
    assert D[v0] == 1 and D[v0 + 17] == 1 and oob.value == 0
    assert Rbin[0] == v0 and Mu[v0] == 1


def test_oracle_under_sanitizers(O, tmp_path):
    """the CPU restatement is clean under ASan+UBSan on a short chain"""
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include "smc_oracle.h"
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    orc_sys s = {256, 3, 33.0, 240.0, 3.0, 5.960464477539063e-9, 2.44140625e-5, 33, 33};
    double *R = calloc(768, sizeof(double)), W[18];
    uint64_t zh[33];
    orc_chain_result res;
    orc_initialize_walls(1.6, 0.0, 3.0, 0.5, 3, 0.0, W);
    if (orc_fcc_init(4, 4, 33.0, 240.0, R) != 256) return 2;
    if (orc_chain(&s, 12345u, R, W, 1.1, 1.1, 2, 4, 2, ORC_FLAG_E0_RESTART, NULL, NULL, zh, NULL, NULL, &res)) return 3;
    printf("%.17g %llu\n", res.meanE, (unsigned long long)res.accepted);
    free(R);
    return 0;
}
''')
    exe = tmp_path / "drv"
    subprocess.check_call(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-I", O.ORACLE_DIR, str(drv),
                           os.path.join(O.ORACLE_DIR, "smc_oracle.c"), "-lm", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # and the sanitised build agrees with the normal one
    ref = O.chain(O.make_sys(256), 12345, O.fcc(4, 4), O.W_FIXTURE, 1.1, 1.1, 2, 4, 2)
    meanE, acc = out.stdout.split()
    assert float(meanE) == ref["meanE"] and int(acc) == ref["accepted"]


def test_pressure_restatement(O):
    """pressure (SMC.c:696-720) is the virial of the pair energy: P = -(1/3V) sum r dV/dr, i.e.
    the derivative of energy() under a uniform dilation; wallsPressure keeps the reference's
    geometry (distance from z + L/2, SMC.c:880), checked here against an independent numpy form."""
    rs = np.random.RandomState(4)
    N = 96
    s = O.make_sys(N, L=8.0, Lz=8.0, cutoff=3.0)
    R = ((rs.rand(N, 3) - 0.5) * np.array([8.0, 8.0, 4.0])).ravel()
    dp = C.POINTER(C.c_double)
    P = O.lib().orc_pressure(C.byref(s), R.ctypes.data_as(dp))
    # numpy restatement of the same double sum
    X = R.reshape(-1, 3)
    d = X[:, None, :] - X[None, :, :]
    d[..., 0] -= 8.0 * np.rint(d[..., 0] / 8.0); d[..., 1] -= 8.0 * np.rint(d[..., 1] / 8.0)
    r2 = (d ** 2).sum(-1)[np.triu_indices(N, 1)]
    r2 = r2[r2 < 9.0]
    assert abs(P - (-(24.0 / r2 ** 3 - 48.0 / r2 ** 6).sum() / (3 * 8.0 * 8.0 * 8.0))) < 1e-9 * abs(P)
    W = O.W_FIXTURE
    Pw = O.lib().orc_walls_pressure(C.byref(s), R.ctypes.data_as(dp), W.ctypes.data_as(dp))
    acc = 0.0
    dw = 8.0 / 3
    for m in range(9):
        dx = X[:, 0] - (m // 3) * dw; dx -= 8.0 * np.rint(dx / 8.0)
        dy = X[:, 1] - (m % 3) * dw; dy -= 8.0 * np.rint(dy / 8.0)
        dz = X[:, 2] + 8.0 / 2; dz -= 8.0 * np.rint(dz / 8.0)
        q = dx * dx + dy * dy + dz * dz
        k = q < 9.0
        acc += (24.0 * W[2 * m + 1] / q[k] ** 3 - 48.0 * W[2 * m] / q[k] ** 6).sum()
        acc += (24.0 * s.b0 / dz[k] ** 6 - 48.0 * s.a0 / dz[k] ** 12).sum()
    assert abs(Pw - (-acc / (3 * 8.0 * 8.0 * 8.0))) < 1e-9 * abs(Pw)


def test_fft_acf_restatement(O):
    """8f.3: the FFT form of the reference's autocorrelation against its defining sums"""
    rs = np.random.RandomState(8)
    for n, kmax in ((64, 10), (65, 10), (41, 2500000), (200, 2500000)):
        H = -300 + np.cumsum(rs.standard_normal(n))
        a, b = O.fft_acf(H, kmax), O.fft_acf_direct(H, kmax)
        assert len(a) == len(b) == (kmax if n >= 2 * kmax + 1 else n // 2 - 2)
        assert a[0] == 1.0 and np.abs(a - b).max() < 1e-10


# ------------------------------------------------------------------ SURVEY 8f.4
def _cluster_analysis_py(N, r, L, cut):
    """independent pure-Python restatement of clusterAnalysis (SMC.c:971-1045), small N only"""
    r = np.asarray(r, dtype=np.float64).reshape(N, 3)
    npairs = N * (N - 1) // 2
    tri = lambda k: (k * k - 3 * k + 2) // 2          # SMC.c:987
    num1 = np.zeros(npairs + 1, dtype=np.int64)
    num2 = np.zeros(npairs + 1, dtype=np.int64)
    num3 = np.zeros(npairs + 1, dtype=np.int64)
    for l in range(1, N):
        for i in range(l):
            d = r[l] - r[i]
            d[0] -= L * np.rint(d[0] / L)
            d[1] -= L * np.rint(d[1] / L)
            if d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < cut * cut:
                num1[tri(l) + i] = 1
    cn = [0] * 8
    over = 0
    for l in range(1, N):
        for i in range(l):
            idx = tri(l) + i
            if not num1[idx]:
                continue
            for i2 in range(l):
                if i2 == i:
                    continue
                if num1[idx - i + i2] & num1[tri(i2) + i]:
                    if num2[idx] < 8:
                        cn[num2[idx]] = i2
                    else:
                        over += 1
                    num2[idx] += 1
            if num2[idx] > 1:
                for m in range(1, min(num2[idx], 8)):
                    if num1[tri(cn[m]) + cn[m - 1]]:
                        num3[idx] += 1
    return np.stack([num1[:npairs], num2[:npairs], num3[:npairs]], axis=1), over


@pytest.mark.parametrize("N,L,cut,seed", [(48, 6.0, 1.7, 1), (64, 5.0, 1.7, 2), (40, 4.0, 2.2, 3)])
def test_cluster_analysis_restatements_agree(O, N, L, cut, seed):
    rs = np.random.RandomState(seed)
    r = rs.uniform(-L / 2, L / 2, (N, 3))
    r[:, 2] *= 0.6
    got, ov = O.cluster_analysis(N, r.ravel(), L, cut)
    ref, ovr = _cluster_analysis_py(N, r, L, cut)
    assert np.array_equal(got, ref) and ov == ovr
    assert got[:, 0].sum() > N
    if cut > 2:
        assert ov > 0   # the common_nn[8] overflow branch is exercised
    n1, h2, h3 = O.cluster_counts(N, got)
    assert n1 == got[:, 0].sum() and h2.sum() == n1 and h3.sum() == n1


def test_cluster_analysis_shared_entries(O):
    """the reference's pair index makes (l, l-1) and (l+1, 0) share an entry: a bond between
    particles 2 and 1 alone also marks the (3,0) entry, and the entry's counters add up"""
    N, L = 6, 50.0
    r = np.array([[8.0 * k - 20.0, 0.0, 0.0] for k in range(N)])
    r[2] = r[1] + [1.0, 0, 0]          # only (2,1) is inside the cutoff
    LCA, ov = O.cluster_analysis(N, r.ravel(), L, 1.7)
    tri = lambda k: (k * k - 3 * k + 2) // 2
    assert tri(2) + 1 == tri(3) + 0 == 1
    assert LCA[:, 0].tolist() == [0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    assert ov == 0 and LCA[:, 1:].sum() == 0


def test_chain_with_cluster_analysis_cadence(O):
    """k % LCA_TIME == 0 with k = (n+1)/gather_lapse (SMC.c:138, 143)"""
    s = O.make_sys(108, L=6.0)
    R0 = O.fcc(3, 3, L=6.0)
    out = O.chain(s, 7, R0, O.W_FIXTURE, 1.1, 1.1, 0, 12, 2, lca_time=3, lca_cutoff=1.7)
    assert out["gathers"] == 6 and out["lca"]["analyses"] == 2
    assert out["lca"]["n1"] > 0 and out["lca"]["h2"].sum() == out["lca"]["n1"]
