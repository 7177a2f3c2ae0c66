"""CPU tests: the oracle (oracle/smc_oracle.c) against the REAL reference, bit for bit.

oracle/build_ref.sh compiles the hot-path line ranges of /root/reference/SMC.c and
SMC_noMPI_noWall.c where they lie (one library per compile-time N, nothing copied to disk) into
oracle/_ref/.  tests/golden/ref_smc.json holds their outputs on tests/ref_cases.GOLDEN_CASES
(generator: tests/golden/make_ref_golden.py); these tests run the same cases through the oracle
and demand identical bits: K1-K4 values, K5 totals, pressure, whole chains through
oneParticleMoves with sMC's loop (energy series, accepted counts of every sweep, final positions,
full density/mobility histograms, results.E/dE/acceptance_ratio/cv), initializeBox,
initializeWalls, clusterAnalysis, and the noWall variant's sweep (BASELINE config 1).
Where oracle/_ref is present (the build container; it also travels to the GPU box as prebuilt
.so files) a second, larger set is compared live.
"""
import json
import os

import pytest

import ref_cases as RC
import ref_lib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_smc.json")))


def _ident(i):
    c = GOLD["cases"][i]["case"]
    return "%s-N%d-%d" % (c["kind"], c["N"], i)


def _tuplify(c):
    c = dict(c)
    if "start" in c:
        c["start"] = tuple(c["start"])
    return c


def test_golden_file_describes_the_cases_of_this_tree():
    assert [_tuplify(e["case"]) for e in GOLD["cases"]] == [dict(c) for c in RC.GOLDEN_CASES]


def _diff(exp, got, path=""):
    """first differing leaf, for a readable failure"""
    if isinstance(exp, dict):
        for k in exp:
            d = _diff(exp[k], got.get(k), path + "/" + k)
            if d:
                return d
        return None
    if isinstance(exp, list) and isinstance(got, list) and len(exp) == len(got):
        for k, (a, b) in enumerate(zip(exp, got)):
            d = _diff(a, b, "%s[%d]" % (path, k))
            if d:
                return d
        return None
    return None if exp == got else "%s: reference %r, oracle %r" % (path, exp, got)


@pytest.fixture(scope="module")
def oracle_backend():
    return RC.OracleBackend()


@pytest.mark.parametrize("i", range(len(GOLD["cases"])), ids=_ident)
def test_oracle_reproduces_the_real_reference_bit_for_bit(oracle_backend, i):
    entry = GOLD["cases"][i]
    got = RC.compute(oracle_backend, [_tuplify(entry["case"])])[0]
    assert _diff(entry["expect"], got) is None


_have_ref = all(ref_lib.available(n) for n in (108, 256, 1024, 4096)) and ref_lib.available(256, nw=True)


@pytest.mark.skipif(not _have_ref, reason="oracle/_ref not built (no reference tree and no prebuilt libraries)")
@pytest.mark.parametrize("seed0", [1, 2])
def test_oracle_against_the_reference_libraries_live(oracle_backend, seed0):
    cases = RC.live_cases(seed0)
    ref = RC.compute(RC.RefBackend(), cases)
    got = RC.compute(oracle_backend, cases)
    for c, a, b in zip(cases, ref, got):
        assert _diff(a, b) is None, c


@pytest.mark.skipif(not _have_ref, reason="oracle/_ref not built")
def test_reference_libraries_are_the_reference(oracle_backend):
    """the libraries report the macros of SMC.h:26-58 and reproduce what SURVEY 8c recorded from the
    survey-time build of the whole reference: the wall fixture and the 20-sweep N=108 chain"""
    import numpy as np
    import oracle_lib as O
    r = ref_lib.RefSMC(108)
    assert (r.N, r.M, r.ncx, r.ncz, r.cutoff) == (108, 3, 33, 33, 3.0)
    assert (r.a0, r.b0) == (O.A0_PLANE, O.B0_PLANE)
    W = r.initialize_walls()
    assert np.array_equal(W, O.W_FIXTURE)
    X = r.initialize_box(33.0, 200.0)
    E0 = r.energy(X, 33.0) + r.walls_energy(X, W, 33.0, 200.0)
    assert r.walls_energy(X, W, 33.0, 200.0) == -4.0581627260515186e-14
    E, jj = r.sweeps(12345, X, W, 33.0, 200.0, 1.1, 1.1, 20, E0)
    assert E[-1] == -3.8631457699032183 and int(jj.sum()) == 2048


def test_timing_build_of_the_reference_equals_the_oracle_on_one_sweep():
    """bench.py's cpu_baseline leg, kind "reference": oracle/time_ref.py runs the -O3 build of the REAL oneParticleMoves
    (oracle/_ref/libref_smc_N4096_O3.so).  It is a timing leg, but what it times must be the sweep the bench's GPU side runs:
    same start, same seed -> the accepted count of the oracle, the energy to contraction-level rounding."""
    import subprocess
    import sys
    import numpy as np
    import oracle_lib as O
    so = os.path.join(ref_lib.REF_DIR, "libref_smc_N4096_O3.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_smc_N4096_O3.so not built (no reference tree here)")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "oracle", "time_ref.py"), "4096", "8", "16",
                          "12345", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    secs, acc, E1 = out.stdout.split()
    assert float(secs) > 0.0
    s = O.make_sys(4096)
    ch = O.chain(s, 12345, O.fcc(8, 16), O.W_FIXTURE, 1.1, 1.1, 0, 1, 10)
    assert int(acc) == int(ch["jj"][0])
    assert abs(float(E1) - ch["E"][1]) < 1e-9 * max(1.0, abs(ch["E"][1]))


@pytest.mark.skipif(not _have_ref, reason="oracle/_ref not built")
def test_fft_acf_restatement_against_the_references_own_simple_acf():
    """Row 8f.3 has NO reference pin: fft_acf (SMC.c:1055-1093) needs FFTW, which the image lacks, and no stand-in is written.
    The reference's OTHER autocorrelation, simple_acf (SMC.c:1096-1122: plain sums, compiled into oracle/_ref from the
    reference's own text), gives an independent cross-check of the restatement where the two definitions are expected to agree:
    fft_acf transforms the HALF spectrum back with a transform of HALF the length (SMC.c:1063, 1083), so its entry i is the
    circular autocorrelation at lag 2 i (exactly: (n c[2i] - P_nyq) / (n c[0] - P_nyq) for even n), while simple_acf[k] is the
    linear one at lag k over the first n - k_max - 1 products.  For a stationary series with n >> k_max they agree to
    O(k_max / n): fft_acf[i] ~ simple_acf[2 i].  Where they do NOT agree: odd entries of simple_acf have no counterpart, the
    circular wrap and the truncation differ by O(k_max / n), and short series (the GPU test's 33..40 samples) differ at O(1).
    This checks the lag structure and normalisation of the restatement against reference CODE; it is not a bit-level pin."""
    import numpy as np
    import oracle_lib as O
    r = ref_lib.RefSMC(108)
    rs = np.random.RandomState(11)
    n, K = 40000, 24
    for rho in (0.9, 0.6):
        e = rs.standard_normal(n)
        H = np.empty(n)
        H[0] = e[0]
        for i in range(1, n):
            H[i] = rho * H[i - 1] + e[i]
        H -= 150.0
        a = O.fft_acf(H, K)
        b = r.simple_acf(H, 2 * K)
        assert a[0] == 1.0 and b[0] == 1.0
        assert np.abs(a - b[0:2 * K:2]).max() < 0.02, np.abs(a - b[0:2 * K:2]).max()
        # the series really decays over these lags, so a wrong lag map (entry i against lag i) could not pass
        assert np.abs(a[1:8] - b[1:8]).max() > 0.1
        assert abs(a[3] - rho ** 6) < 0.05
