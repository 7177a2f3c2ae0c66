"""CPU tests of the product's host side: the C-ABI library loads, exports every
symbol include/smcx.h declares, validates its arguments, fails loudly without a
GPU (no CPU fallback), and its srand() state agrees with the oracle's rand().
No compute entry is exercised here (there is no GPU in this container)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "smcx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(smcx_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(S):
    lib = C.CDLL(S.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsmcx.so does not export %s" % n
    assert sorted(S.EXPORTS) == names  # the Python binding covers the whole ABI


def test_roofline_refuses_instruction_counts_of_another_build(S, tmp_path, monkeypatch):
    """bench.py's roofline fraction comes from committed PMC instruction counts; they carry the source identity of the library
    they were taken from (smcx_kernel_source_id: sha256 of the generated body / the source files) and are used only for the
    same build of the kernel -- a kernel edited without a re-profile yields frac: null with a note, not a stale number."""
    import importlib
    import json
    kn = "smcx::sweep_kernel_mc64"
    sid = S.kernel_source_id(kn)
    assert sid and re.fullmatch(r"[0-9a-f]{16}", sid)
    assert S.kernel_source_id("smcx::sweep_kernel_mt64x8") not in (None, sid)
    # names this library does not build -- retired kernels, typos, a bare prefix -- have NO id (smcx.h: SMCX_ERR_PARAM)
    for unknown in ("smcx::sweep_kernel_mt16x2", "smcx::sweep_kernel_mt32x16", "smcx::sweep_kernel_mc128", "smcx::sweep_kernel",
                    "smcx::sweep_kernel_mi", "smcx::sweep_kernel_mc64 "):
        assert S.kernel_source_id(unknown) is None, unknown
    assert S.kernel_source_id("smcx::sweep_kernel<8, 2, 4, 2>") == S.kernel_source_id("smcx::sweep_kernel_lead<16, 4, 3, 2>")
    assert S.kernel_source_id("smcx::sweep_kernel_mx<64, 1, 4, true>")
    assert S.kernel_source_id("smcx::sweep_kernel_mi<64, 4, 4>") and S.kernel_source_id("smcx::sweep_kernel_lead<16, 4, 3, 2>")
    assert S.kernel_source_id("smcx::no_such_kernel") is None
    bench = importlib.import_module("bench")
    (tmp_path / "profiles").mkdir()
    entry = {"workload": {"N": 4096, "replicas": 4096, "sweeps_in_launch": 1, "start": "fcc(8,16)"}, "source_id": sid,
             "per_wave_move": {"SQ_INSTS_VALU": 300.0, "SQ_INSTS_VALU_ADD_F64": 30.0, "SQ_INSTS_VALU_MUL_F64": 30.0,
                               "SQ_INSTS_VALU_FMA_F64": 30.0, "SQ_INSTS_VALU_TRANS_F64": 2.0, "SQ_INSTS_SALU": 100.0}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    for ident, ok in ((sid, True), ("0123456789abcdef", False), (None, False)):
        entry["source_id"] = ident
        json.dump({kn: entry}, open(tmp_path / "profiles" / "kernel_counters.json", "w"))
        r = bench.issue_roofline(kn, 10.0, 4096, 4096, 2.35, start="fcc(8,16)")
        if ok:
            assert 0.1 < r["frac"] < 1.0 and r["valu_per_move"] == 300.0
        else:
            assert r["frac"] is None and "another build" in r["note"]


def test_window_schedule_runs_every_sweep_of_every_replica_once_and_in_order(S):
    """smcx_debug_window_schedule = the schedule launch_sweeps_* follow (csrc/smcx_sweep_ma.hip: unit_launches).  A launch group of
    `nsweeps` sweeps is cut into blocks of `every`; with a replica count that is no multiple of what the device holds the group runs
    as windows of `granule` consecutive (replica, block) units.  Whatever the numbers: every replica runs every sweep exactly once
    and in order, never twice in one launch (a replica's sweeps are sequential: SMC.c:292-347), no window holds more than
    `granule` workgroups, and there are ceil(nrep x blocks / granule) of them; multiples of the granule and counts below it keep
    the plain launches (one per block over all replicas)."""
    rs = np.random.RandomState(3)
    cases = [(4097, 4096, 9, 2), (4097, 4096, 1, 2), (6144, 4096, 10, 2), (9000, 4096, 11, 2), (261, 256, 7, 1), (300, 256, 1, 1),
             (5121, 5120, 16, 2), (4096, 4096, 9, 2), (8192, 4096, 10, 2), (100, 4096, 5, 2), (7, 3, 5, 3), (5, 4, 4, 1)]
    cases += [(int(rs.randint(1, 3000)), int(rs.randint(1, 700)), int(rs.randint(0, 17)), int(rs.randint(1, 5))) for _ in range(60)]
    for nrep, G, nsw, every in cases:
        rows, n = S.window_schedule(nrep, G, nsw, every)
        assert n == len(rows)
        nb = (nsw + every - 1) // every
        windows = nrep > G and nrep % G != 0
        done = np.zeros(nrep, dtype=np.int64)            # sweeps each replica has run so far
        for grid, u0, nmod, blk0, use, sa, na, sb, nb_ in rows:
            assert bool(use) == windows and (nmod == nrep if windows else nmod == 0), (nrep, G, nsw, every)
            assert grid <= (G if windows else nrep)
            u = u0 + np.arange(grid)
            rep = u % nrep if windows else np.arange(grid)
            blk = u // nrep if windows else np.full(grid, blk0)
            assert len(np.unique(rep)) == grid                      # never twice in one launch
            assert np.all((blk == blk0) | (blk == blk0 + 1))
            sw0 = np.where(blk == blk0, sa, sb); cnt = np.where(blk == blk0, na, nb_)
            assert np.all(sw0 == blk * every) and np.all(cnt == np.minimum(every, nsw - blk * every)) and np.all(cnt > 0)
            assert np.all(done[rep] == sw0)                         # in order: this block starts where the replica stands
            done[rep] += cnt
        assert np.all(done == nsw), (nrep, G, nsw, every)
        assert n == (-(-nrep * nb // G) if windows else nb), (nrep, G, nsw, every, n)
    with pytest.raises(S.SmcxError):
        S.window_schedule(0, 4096, 4, 2)


def test_host_library_exports_every_declared_symbol(S):
    """libsmcx_host.so (plain C above the ABI, incl. the RCCL multi-GPU driver) against include/smcx_host.h"""
    hdr = open(os.path.join(ROOT, "include", "smcx_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(smcx_host_[a-z_0-9A-Z]+)\s*\(", hdr)))
    assert "smcx_host_sMC_multi" in names and len(names) >= 12
    lib = C.CDLL(S.HOST_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libsmcx_host.so does not export %s" % n
    assert sorted(S.HOST_EXPORTS) == names
    out = subprocess.run(["ldd", S.HOST_LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in out and "libsmcx.so" in out and "oracle" not in out
    assert C.CDLL(S.LIB_PATH).smcx_strerror  # (status 7 = the RCCL gather failed)
    f = C.CDLL(S.LIB_PATH).smcx_strerror
    f.restype = C.c_char_p
    assert f(S.ERR_RCCL) == b"RCCL collective failed"


def test_nowall_host_path_equals_the_real_reference(S):
    """BASELINE config 1 (SMC_noMPI_noWall.c, one chain on the host CPU) through the C host library against the
    outputs of the REAL file (oracle/_ref/libref_nw_N*.so -> tests/golden/ref_smc.json, cases "nw"): start lattice,
    E0, P0, energySingle/force of three particles, accepted moves of every sweep, energy and positions after every
    sweep, final pressure -- bit for bit."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_smc.json")))["cases"]
    cases = [e for e in gold if e["case"]["kind"] == "nw"]
    assert len(cases) >= 4
    dig = lambda a: "%s:%s" % (a.dtype.str, hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())
    for e in cases:
        c, exp = e["case"], e["expect"]
        nw = S.NoWall(c["N"], c["rho"])
        assert float(nw.L).hex() == exp["L"]
        X, placed = nw.fcc()
        assert placed == c["N"] and dig(X) == exp["X"]
        assert float(nw.energy(X)).hex() == exp["E0"] and float(nw.pressure(X)).hex() == exp["P0"]
        for i, want in zip((0, 1, c["N"] - 1), exp["single"]):
            en, F = nw.single(X, i)
            assert [float(v).hex() for v in (en, *F)] == want, (c, i)
        # sweep by sweep: the energy after sweep n is E[k] of the NEXT gather with gather_lapse = 1
        R = X.copy()
        E, P, jj = nw.sMC(R, c["T"], c["A"], c["seed"], c["steps"], 1)
        assert [int(v) for v in jj] == exp["jj"], c
        assert float(E[0]).hex() == exp["E0"]
        assert [float(v).hex() for v in E[1:]] == exp["Es"][:-1], c      # E[k] is taken BEFORE sweep k
        assert dig(R) == exp["Rs"][-1] and float(nw.energy(R)).hex() == exp["Es"][-1]
        assert float(nw.pressure(R)).hex() == exp["P_end"]
        # a shorter chain lands on the intermediate positions
        R2 = X.copy()
        nw.sMC(R2, c["T"], c["A"], c["seed"], 3, 1)
        assert dig(R2) == exp["Rs"][2]
    # the command-line form with the reference main's own parameters (rho 0.1, T 0.4, A 4e-8) and ITS box:
    # L = cbrt(N/rho) by this libc's cbrt, as the reference's main computes it (:82) -- one ulp below numpy's here,
    # which moves pairs across the cutoff L/2 of the perfect lattice (E0 = -36.1924 instead of SURVEY 8c's -36.1886)
    box = S._host().smcx_host_nowall_box
    box.restype, box.argtypes = C.c_double, [C.c_int, C.c_double]
    nw = S.NoWall(256, 0.1)
    nw.L = box(256, 0.1)
    X, _ = nw.fcc()
    exe = os.path.join(os.path.dirname(S.LIB_PATH), "smcx_main")
    r = subprocess.run([exe, "--nowall", "10", "2", "256", "12345"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and ("E[0] = %0.12f" % nw.energy(X)) in r.stdout, r.stdout + r.stderr


def test_no_oracle_in_product():
    """the product never links or imports the oracle"""
    pkg = os.path.join(ROOT, "montecarlo-surfacer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("oracle/ is", ""), os.path.join(dirpath, f)
    out = subprocess.run(["ldd", os.path.join(pkg, "libsmcx.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out


def test_default_params_are_the_reference_macros(S):
    p = S.default_params(108, 2)
    assert (p.N, p.M, p.nrep) == (108, 3, 2)
    assert (p.L, p.Lz, p.T, p.A, p.cutoff) == (33.0, 240.0, 1.1, 1.1, 3.0)   # main.c:41-51, SMC.h:38
    assert (p.a0, p.b0) == (5.960464477539063e-9, 2.44140625e-5)            # SMC.h:32-33
    assert (p.Ncx, p.Ncz) == (33, 33)                                       # SMC.h:53-55
    assert p.flags == S.FLAGS_REFERENCE and p.base_seed == 12345


def test_parameter_validation_and_loud_failure_without_gpu(S):
    lib = S._lib()
    h = C.c_void_p()
    for kw, want in ((dict(N=107), S.ERR_PARAM), (dict(N=0), S.ERR_PARAM), (dict(nrep=0), S.ERR_PARAM),
                     (dict(L=-1.0), S.ERR_PARAM), (dict(T=0.0), S.ERR_PARAM), (dict(Ncz=300), S.ERR_PARAM),
                     (dict(M=6), S.ERR_UNSUPPORTED), (dict(M=0), S.ERR_PARAM),
                     (dict(tune_kernel=9), S.ERR_PARAM), (dict(tune_kernel=-1), S.ERR_PARAM),
                     (dict(tune_resort=-1), S.ERR_PARAM),
                     (dict(flags=S.FLAGS_REFERENCE | S.FLAG_CLUSTERS, lca_time=0), S.ERR_PARAM),
                     (dict(flags=S.FLAGS_REFERENCE | S.FLAG_CLUSTERS, lca_cutoff=0.0), S.ERR_PARAM)):
        p = S.default_params(108, 2)
        for k, v in kw.items():
            setattr(p, k, v)
        assert lib.smcx_create(C.byref(p), C.byref(h)) == want, kw
        assert not h.value
    if S.device_count() == 0:
        p = S.default_params(108, 2)
        assert lib.smcx_create(C.byref(p), C.byref(h)) == S.ERR_NODEVICE
        with pytest.raises(S.SmcxError) as e:
            S.Engine(p)
        assert e.value.status == S.ERR_NODEVICE
        with pytest.raises(S.SmcxError):
            S.eval_moves(p, np.zeros((2, 324)), np.zeros(18), np.zeros(2, dtype=np.int32), np.zeros((2, 3)))
    assert lib.smcx_strerror(S.ERR_NODEVICE) == b"no HIP device"
    assert lib.smcx_destroy(None) == S.OK


def test_stale_tuning_variable_is_refused():
    """the sweep kernels' measurement switches are smcx_params fields; an SMCX_* switch left in the environment makes
    smcx_create fail (before it even looks for a device).  The product library never reads them, whatever
    SMCX_ALLOW_ENV_TUNING says: only VARIANT builds (make VARIANT=x, A/B sessions) do."""
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); import smcx_loader; S = smcx_loader.load(); "
            "p = S.default_params(4096, 4); h = C.c_void_p(); rc = S._lib().smcx_create(C.byref(p), C.byref(h)); "
            "print(rc, S._lib().smcx_last_error_string(None).decode())" % ROOT)
    for var in ("SMCX_MB", "SMCX_MC", "SMCX_MA", "SMCX_MI", "SMCX_MX", "SMCX_RESORT", "SMCX_NO_LEAD"):
        for allow in (False, True):
            env = {k: v for k, v in os.environ.items() if not k.startswith("SMCX_")}
            env[var] = "0"
            if allow:
                env["SMCX_ALLOW_ENV_TUNING"] = "1"
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-1500:]
            rc, msg = r.stdout.strip().split(" ", 1)
            assert int(rc) == 1 and var in msg and "VARIANT" in msg, r.stdout


def _emulated_screen(S, p, lds_z, P, X):
    """the screen of csrc/smcx_sweep_mx.hip (mx_qxy, mx_screen / mx_screen_lds) in numpy with the
    device's roundings: int16 wrap of x,y, saturating dot2, fp32 fma, fp32 or fp16 z.  P: probes
    [n,3], X: particles [n,3]; returns (candidate flag, exact in-cutoff flag, unsafe flag)"""
    thr, u2, to_fixed, zsafe = S.screen_bound(p, lds_z)
    f32 = np.float32

    def pack(v):   # mx_pack_xy: rint, then the low 16 bits as int16
        return np.rint(v * to_fixed).astype(np.int64).astype(np.uint16).astype(np.int16)

    dxi = (pack(P[:, 0]).astype(np.int32) - pack(X[:, 0]).astype(np.int32)).astype(np.int16).astype(np.int64)  # v_pk_sub_i16 wraps
    dyi = (pack(P[:, 1]).astype(np.int32) - pack(X[:, 1]).astype(np.int32)).astype(np.int16).astype(np.int64)
    i2 = np.minimum(dxi * dxi + dyi * dyi, 2 ** 31 - 1)                      # v_dot2_i32_i16 clamp
    t = (f32(u2).astype(np.float64) * i2.astype(f32).astype(np.float64) - f32(thr).astype(np.float64)).astype(f32)  # v_fma_f32
    if lds_z:
        zh, ph = X[:, 2].astype(f32).astype(np.float16), P[:, 2].astype(f32).astype(np.float16)
        dz = (ph.astype(f32) - zh.astype(f32)).astype(np.float16).astype(np.float64)  # v_pk_add_f16
    else:
        dz = (P[:, 2].astype(f32) - X[:, 2].astype(f32)).astype(np.float64)            # v_sub_f32 (exact in double, then f32)
        dz = dz.astype(f32).astype(np.float64)
    q = (dz * dz + t.astype(np.float64)).astype(f32)                                   # v_fma(_mix)_f32
    cand = np.signbit(q)
    unsafe = ~(np.abs(X[:, 2]) < zsafe) | ~(np.abs(P[:, 2]) < zsafe)
    d = P - X
    d[:, 0] -= p.L * np.rint(d[:, 0] / p.L); d[:, 1] -= p.L * np.rint(d[:, 1] / p.L)
    exact = (d * d).sum(axis=1) < p.cutoff ** 2
    return cand, exact, unsafe


@pytest.mark.parametrize("L,Lz,lds_z", [(33.0, 240.0, 0), (33.0, 240.0, 1), (6.5, 240.0, 1), (100.0, 60.0, 0),
                                        (100.0, 900.0, 1)])
def test_screen_never_misses_a_pair_inside_the_cutoff(S, L, Lz, lds_z):
    """the proof obligation of the screened sweep kernel, checked on the CPU with the
    product's own threshold (smcx_screen_bound) and the device's arithmetic emulated: pairs placed on
    both sides of the cutoff sphere, also across the periodic x,y edges and at the edge of the safe
    z range; every pair inside the cutoff must be a candidate (or sit in the unsafe set, which the
    kernel always passes on), and the candidates must stay close to the true hits"""
    p = S.default_params(1024, 1, L=L, Lz=Lz)
    rs = np.random.RandomState(int(L * 10 + Lz) + lds_z)
    n = 2_000_000
    thr, u2, to_fixed, zsafe = S.screen_bound(p, lds_z)
    assert thr > p.cutoff ** 2 and zsafe >= Lz / 2
    X = np.empty((n, 3))
    X[:, 0] = rs.uniform(-L / 2, L / 2, n); X[:, 1] = rs.uniform(-L / 2, L / 2, n)
    X[:, 2] = rs.uniform(-1.05, 1.05, n) * min(zsafe, 4 * Lz)
    X[: n // 8, 2] = np.sign(X[: n // 8, 2]) * zsafe * (1 - 10.0 ** rs.uniform(-7, -1, n // 8))   # just inside the safe range
    r = p.cutoff * (1 + rs.uniform(-1, 1, n) * 10.0 ** rs.uniform(-9, -0.5, n))                # radii around the cutoff
    r[: n // 16] = rs.uniform(0.3, p.cutoff, n // 16)
    v = rs.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
    P = X + r[:, None] * v
    P[:, 0] -= L * np.rint(P[:, 0] / L); P[:, 1] -= L * np.rint(P[:, 1] / L)                 # probes are kept wrapped
    cand, exact, unsafe = _emulated_screen(S, p, lds_z, P, X)
    assert exact.sum() > n // 4 and (~exact).sum() > n // 4
    missed = exact & ~cand & ~unsafe
    assert not missed.any(), (missed.sum(), P[missed][:3], X[missed][:3])
    # and it is not vacuous: pairs clearly outside are rejected (the roundings that the margin covers can
    # also make a pair look closer, so a false candidate is at most about two margins outside)
    d2 = ((P - X) ** 2).sum(axis=1)  # not minimum image: only used far from the edges below
    inner = (np.abs(X[:, 0]) < L / 2 - 2 * p.cutoff) & (np.abs(X[:, 1]) < L / 2 - 2 * p.cutoff) & ~unsafe
    if inner.sum() > 1000:
        assert not (cand & inner & (d2 > p.cutoff ** 2 + 2.2 * (thr - p.cutoff ** 2) + 0.01)).any()


def test_host_c_system_preparation_matches_reference_fixtures(S, O):
    """host/smcx_host.c against the reference's own outputs and the oracle:
    initializeWalls (SMC.c:475-501) gives the 18-value wall fixture the real reference printed, bit
    for bit; the lattices of initializeBox (SMC.c:413-465) and fcc(Na,Nz) equal the oracle's; the
    sizes the reference's rule cannot place (1024: 16 particles, 4096: 96, 8192: 128 left at the
    origin, SMC.c:416-431) are reported as such."""
    W = S.initialize_walls()                                   # M=3, (1.6, 0.0, 3.0, 0.5), uninit = 0.0
    assert np.array_equal(W, O.W_FIXTURE) and np.array_equal(S.W_REFERENCE, O.W_FIXTURE)
    assert np.array_equal(W, O.walls())
    # the ninth site reads uninitialised memory in the reference (SMC.c:481-485): the stand-in value is explicit
    W7 = S.initialize_walls(uninit=0.25)
    assert np.array_equal(W7[:16], W[:16]) and not np.array_equal(W7[16:], W[16:])
    assert np.array_equal(W7, O.walls(uninit=0.25))
    W2 = S.initialize_walls(M=2)                               # even M*M: every site written
    assert W2.shape == (8,) and np.array_equal(W2, O.walls(M=2)) and np.all(W2 > 0)
    for Na, Nz in ((4, 4), (8, 4), (8, 16), (16, 4), (16, 16), (3, 3)):
        assert np.array_equal(S.fcc_init(Na, Nz), O.fcc(Na, Nz)), (Na, Nz)
    assert np.array_equal(S.fcc_init(4, 4, L=20.0, Lz=120.0), O.fcc(4, 4, L=20.0, Lz=120.0))
    for N, L, Lz, unplaced in ((32, 20.0, 120.0, 0), (108, 33.0, 200.0, 0), (256, 33.0, 240.0, 0), (500, 33.0, 240.0, 0),
                               (864, 33.0, 240.0, 0), (2048, 33.0, 240.0, 0), (4000, 33.0, 240.0, 0),
                               (16384, 33.0, 240.0, 0), (1024, 33.0, 240.0, 16), (4096, 33.0, 240.0, 96),
                               (8192, 33.0, 240.0, 128)):
        X, placed = S.initialize_box(N, L, Lz)
        Xo, placed_o = O.box_ref(N, L, Lz)
        assert placed == placed_o == N - unplaced, (N, placed, placed_o)
        assert np.array_equal(X, Xo), N
        assert np.abs(X[0::3]).max() <= L / 2 and np.abs(X[2::3]).max() <= 0.95 * Lz / 2 + 1e-12
    # N=256 with the reference's rule is the fcc(4,4) start of the benchmark configs
    assert np.array_equal(S.initialize_box(256, 33.0, 240.0)[0], S.fcc_init(4, 4))
    assert np.array_equal(S.initialize_box(16384, 33.0, 240.0)[0], S.fcc_init(16, 16))


def _emulated_int_screen(S, p, P, X):
    """the integer screen of csrc/smcx_sweep_mi.hip (mi_screen4) in numpy with the device's arithmetic:
    int16 wrap of the packed x,y difference INCLUDING the borrow of the 32-bit subtraction into y,
    v_dot2_i32_i16 with the accumulator -C, arithmetic shift, saturating int16 z difference,
    v_mad_i32_i16, sign bit.  Returns (candidate flag, exact in-cutoff flag, unsafe flag)"""
    thr, u2, to_fixed, zsafe, uz, neg_c, zs = S.screen_bound_int(p)

    def pack(v):   # mi_pack_xy: rint, then the low 16 bits
        return np.rint(v * to_fixed).astype(np.int64) & 0xffff

    def word(A):
        return (pack(A[:, 0]) | (pack(A[:, 1]) << 16)).astype(np.uint64)

    def z16(z):    # mi_z16: rint, clamp to +-32767 (v_cvt_i32_f64 saturates), low 16 bits as int16
        return np.clip(np.rint(z / uz), -32767, 32767).astype(np.int64)

    d = (word(P) - word(X)) & np.uint64(0xffffffff)                       # v_sub_u32: the borrow reaches the y field
    dx = (d & np.uint64(0xffff)).astype(np.int64); dx = np.where(dx >= 32768, dx - 65536, dx)
    dy = (d >> np.uint64(16)).astype(np.int64); dy = np.where(dy >= 32768, dy - 65536, dy)
    I = dx * dx + dy * dy + neg_c                                           # v_dot2_i32_i16, no clamp: must fit int32
    assert I.max() < 2 ** 31 and I.min() >= -2 ** 31
    I = I >> (2 * zs)                                                       # v_ashrrev_i32
    dz = np.clip(z16(P[:, 2]) - z16(X[:, 2]), -32768, 32767)               # v_pk_sub_i16 clamp
    q = dz * dz + I                                                         # v_mad_i32_i16
    assert q.max() < 2 ** 31
    cand = q < 0
    unsafe = ~(np.abs(X[:, 2]) < zsafe) | ~(np.abs(P[:, 2]) < zsafe)
    dd = P - X
    dd[:, 0] -= p.L * np.rint(dd[:, 0] / p.L); dd[:, 1] -= p.L * np.rint(dd[:, 1] / p.L)
    exact = (dd * dd).sum(axis=1) < p.cutoff ** 2
    return cand, exact, unsafe, thr


@pytest.mark.parametrize("L,Lz", [(33.0, 240.0), (33.0, 200.0), (6.5, 240.0), (100.0, 60.0), (100.0, 900.0), (16.0, 240.0)])
def test_integer_screen_never_misses_a_pair_inside_the_cutoff(S, L, Lz):
    """the proof obligation of sweep_kernel_mi's all-integer screen, checked on the CPU with the product's
    own numbers (smcx_screen_bound_int): 2e6 pairs per box placed within 1e-9 .. 0.3 of the cutoff sphere,
    across the periodic x,y edges and up to the edge of the safe z range.  Every pair inside the cutoff must
    be flagged (or sit in the unsafe set, which the kernel always passes on); no intermediate leaves int32;
    and the screen stays tight: nothing further out than the threshold plus its own margin is flagged."""
    p = S.default_params(1024, 1, L=L, Lz=Lz)
    rs = np.random.RandomState(int(L * 10 + Lz))
    n = 2_000_000
    thr, u2, to_fixed, zsafe, uz, neg_c, zs = S.screen_bound_int(p)
    assert thr > p.cutoff ** 2 and thr < 1.02 * p.cutoff ** 2 + 40 * uz and zsafe >= 0.55 * Lz and zs in (4, 6)
    X = np.empty((n, 3))
    X[:, 0] = rs.uniform(-L / 2, L / 2, n); X[:, 1] = rs.uniform(-L / 2, L / 2, n)
    X[:, 2] = rs.uniform(-1.05, 1.05, n) * zsafe
    X[: n // 8, 2] = np.sign(X[: n // 8, 2]) * zsafe * (1 - 10.0 ** rs.uniform(-7, -1, n // 8))   # just inside the safe range
    r = p.cutoff * (1 + rs.uniform(-1, 1, n) * 10.0 ** rs.uniform(-9, -0.5, n))                # radii around the cutoff
    r[: n // 16] = rs.uniform(0.3, p.cutoff, n // 16)
    v = rs.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
    v[n // 2: n // 2 + n // 8, 2] *= 0.02; v /= np.linalg.norm(v, axis=1)[:, None]               # nearly in-plane pairs
    Pp = X + r[:, None] * v
    Pp[:, 0] -= L * np.rint(Pp[:, 0] / L); Pp[:, 1] -= L * np.rint(Pp[:, 1] / L)                 # probes are kept wrapped
    cand, exact, unsafe, thr = _emulated_int_screen(S, p, Pp, X)
    assert exact.sum() > n // 4 and (~exact).sum() > n // 4
    missed = exact & ~cand & ~unsafe
    assert not missed.any(), (missed.sum(), Pp[missed][:3], X[missed][:3])
    d2 = ((Pp - X) ** 2).sum(axis=1)  # not minimum image: only used far from the edges below
    inner = (np.abs(X[:, 0]) < L / 2 - 2 * p.cutoff) & (np.abs(X[:, 1]) < L / 2 - 2 * p.cutoff) & ~unsafe
    if inner.sum() > 1000:
        assert not (cand & inner & (d2 > p.cutoff ** 2 + 2.2 * (thr - p.cutoff ** 2) + 0.01)).any()
    # far pairs, also further apart in z than the int16 range spans: never flagged by wrap-around
    far = rs.uniform(-1, 1, (200000, 3)) * [L / 2, L / 2, 0.999 * zsafe]
    far2 = rs.uniform(-1, 1, (200000, 3)) * [L / 2, L / 2, 0.999 * zsafe]
    c2, e2, u2_, _ = _emulated_int_screen(S, p, far, far2)
    assert not (e2 & ~c2).any() and (c2 & ~e2).sum() <= 0.02 * max(e2.sum(), 50) + 20


def test_integer_screen_unsupported_boxes_fall_back(S):
    """no built z unit covers a very tall box, and a box narrower than ~2 cutoffs overflows the int32 sum:
    smcx_screen_bound_int says so (the engine then launches the older screened kernel)"""
    for kw in (dict(L=33.0, Lz=5000.0), dict(L=5.0, Lz=240.0, cutoff=3.0)):
        p = S.default_params(1024, 1, **kw)
        with pytest.raises(S.SmcxError) as e:
            S.screen_bound_int(p)
        assert e.value.status == S.ERR_UNSUPPORTED


def _emulated_byte_screen(S, p, P, X):
    """the byte screen of sweep_kernel_mc64 (gen_sweep_ma.py, screen_group8) in numpy with the device's
    arithmetic: one word per particle (z int16 | x int8 << 16 | y int8 << 24, units of L/256, rint), the 32-bit
    subtraction with its borrows from z into x and from x into y, v_dot4_i32_i8 of the difference with itself
    (four signed bytes: low and high byte of dz, dx, dy) on the accumulator -T, sign bit.
    Returns (candidate flag, exact in-cutoff flag, unsafe flag, |dz| in units)"""
    to_fixed, zsafe, neg_t, reach_z = S.screen_bound_byte(p)

    def word(A):
        x = np.rint(A[:, 0] * to_fixed).astype(np.int64) & 0xff
        y = np.rint(A[:, 1] * to_fixed).astype(np.int64) & 0xff
        z = np.clip(np.rint(A[:, 2] * to_fixed), -32767, 32767).astype(np.int64) & 0xffff
        return (z | (x << 16) | (y << 24)).astype(np.uint64), np.clip(np.rint(A[:, 2] * to_fixed), -32767, 32767)

    wp, zp = word(P)
    wx, zx = word(X)
    d = (wp - wx) & np.uint64(0xffffffff)
    acc = np.full(len(d), neg_t, dtype=np.int64)
    for k in range(4):
        b = ((d >> np.uint64(8 * k)) & np.uint64(0xff)).astype(np.int64)
        b = np.where(b >= 128, b - 256, b)
        acc += b * b
    cand = acc < 0
    unsafe = ~(np.abs(X[:, 2]) < zsafe) | ~(np.abs(P[:, 2]) < zsafe)
    dd = P - X
    dd[:, 0] -= p.L * np.rint(dd[:, 0] / p.L); dd[:, 1] -= p.L * np.rint(dd[:, 1] / p.L)
    exact = (dd * dd).sum(axis=1) < p.cutoff ** 2
    return cand, exact, unsafe, np.abs(zp - zx)


@pytest.mark.parametrize("L,Lz", [(33.0, 240.0), (33.0, 60.0), (6.5, 240.0), (12.0, 700.0), (47.9, 240.0)])
def test_byte_screen_never_misses_a_pair_inside_the_cutoff(S, L, Lz):
    """the proof obligation of sweep_kernel_mc64's screen, checked on the CPU with the product's own numbers
    (smcx_screen_bound_byte): 2e6 pairs per box placed within 1e-9 .. 0.3 of the cutoff sphere, across the
    periodic x,y edges and over the whole z range.  Every pair inside the cutoff must be flagged, and its
    |dz| in units must stay within reach_z (the amount the kernel widens a group's z range by: a group out of
    reach holds no neighbour).  The screen stays useful: the flagged volume is below 1.7x the cutoff sphere."""
    p = S.default_params(4096, 1, L=L, Lz=Lz)
    rs = np.random.RandomState(int(L * 10 + Lz))
    n = 2_000_000
    to_fixed, zsafe, neg_t, reach_z = S.screen_bound_byte(p)
    u = 1.0 / to_fixed
    assert zsafe > Lz / 2 and -neg_t < 127 * 127 and (reach_z - 1) * u >= p.cutoff
    X = np.empty((n, 3))
    X[:, 0] = rs.uniform(-L / 2, L / 2, n); X[:, 1] = rs.uniform(-L / 2, L / 2, n)
    X[:, 2] = rs.uniform(-Lz / 2, Lz / 2, n)
    v = rs.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
    r = p.cutoff * (1.0 + rs.choice([-1, 1], n) * 10.0 ** rs.uniform(-9, np.log10(0.3), n))
    Pp = X + v * r[:, None]
    Pp[:, 0] -= L * np.rint(Pp[:, 0] / L); Pp[:, 1] -= L * np.rint(Pp[:, 1] / L)
    cand, exact, unsafe, dzu = _emulated_byte_screen(S, p, Pp, X)
    assert exact.sum() > n // 4 and (~exact).sum() > n // 4
    missed = exact & ~cand & ~unsafe
    assert not missed.any(), (missed.sum(), Pp[missed][:3], X[missed][:3])
    assert dzu[exact].max() <= reach_z
    # tightness: uniformly placed pairs are flagged about as often as the volume ratio of the two spheres says
    # (|dz| kept below 128 units: further apart the low byte of dz aliases -- cells the z ranges keep out of the
    # passes, and which cost a wasted evaluation at worst)
    zs = min(6.0, 60 * u)
    far = rs.uniform(-1, 1, (400000, 3)) * [L / 2, L / 2, zs]
    far2 = rs.uniform(-1, 1, (400000, 3)) * [L / 2, L / 2, zs]
    c2, e2, _, _ = _emulated_byte_screen(S, p, far, far2)
    assert not (e2 & ~c2).any() and c2.sum() <= 1.7 * e2.sum() + 50


def test_byte_screen_unsupported_boxes(S):
    """a box whose L/256 does not resolve the cutoff, or one narrower than two cutoffs: the engine keeps
    sweep_kernel_mb64 / sweep_kernel_mi (int16 units of L/65536)"""
    for kw in (dict(L=100.0, Lz=240.0), dict(L=5.0, Lz=240.0, cutoff=3.0)):
        p = S.default_params(4096, 1, **kw)
        with pytest.raises(S.SmcxError) as e:
            S.screen_bound_byte(p)
        assert e.value.status == S.ERR_UNSUPPORTED


def test_srand_state_agrees_with_oracle_rand(S, O):
    """smcx_rng_seed = srand(): continuing r[i] = r[i-31] + r[i-3] from the exported
    state must give rand()'s outputs."""
    for seed in (0, 1, 42, 12345, 12345 + 4095, 2 ** 31 + 5, 2 ** 32 - 1):
        st = S.rng_seed(seed)
        assert st[31] == 0
        h = [int(v) for v in st[:31]]
        got = []
        for _ in range(200):
            nxt = (h[-31] + h[-3]) & 0xFFFFFFFF
            h.append(nxt)
            got.append(nxt >> 1)
        assert got == list(O.Rng(seed).draws(200)), seed


def test_block_generator_algebra(O):
    """the kernel's 31-at-a-time rand() step (three stride-3 prefix sums, smcx_device.hpp
    rand_block) restated in numpy against the oracle's one-at-a-time generator"""
    def block(h):
        w = h.copy()
        w[:3] = (w[:3] + h[28:31]) & 0xFFFFFFFF
        d = 3
        while d <= 24:
            sh = np.concatenate([np.zeros(d, dtype=np.uint64), w[:-d]])
            w = (w + sh) & 0xFFFFFFFF
            d *= 2
        return w
    import smcx_loader
    S = smcx_loader.load()
    h = S.rng_seed(777)[:31].astype(np.uint64)
    r = O.Rng(777)
    for _ in range(40):
        h = block(h)
        assert list(h >> 1) == list(r.draws(31))


def test_shard_rule():
    sys.path.insert(0, os.path.join(ROOT, "montecarlo-surfacer_amd"))
    import importlib
    dist = importlib.import_module("dist")
    for total, world in ((32768, 8), (10, 4), (3, 8), (4096, 1)):
        got = [dist.shard(total, r, world) for r in range(world)]
        assert sum(c for _, c in got) == total
        pos = 0
        for first, count in got:
            assert first == pos
            pos += count
    assert dist.shard(32768, 3, 8) == (3 * 4096, 4096)
    with pytest.raises(ValueError):
        dist.shard(8, 8, 8)


_GLOO_WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import dist as D
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total, Ncz = 7, 5
first, count = D.shard(total, rank, world)
rec = np.zeros((count, 8)); zh = np.zeros((count, Ncz))
for r in range(count):
    g = first + r
    rec[r] = [100 + g, 11, -5.0 * g, 25.0 * g * g, -g, 0, 10, 0]
    zh[r] = np.arange(Ncz) + g
packed = torch.from_numpy(np.concatenate([rec.ravel(), zh.ravel()]))
obs = D.gather_observables(packed, count, Ncz)
assert obs["accepted"].tolist() == [100 + g for g in range(total)], obs["accepted"]
assert obs["zhist"].shape == (total, Ncz) and obs["zhist"][5, 2] == 7
s = D.summarise(obs, N=10, maxsteps=10)
assert abs(s["mean_of_meanE"] - np.mean([-5.0 * g / 11 for g in range(total)])) < 1e-12
assert np.allclose(s["zprofile"], (np.arange(Ncz) * total + sum(range(total))) / (10 * total))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_gather(tmp_path):
    """the N>1 path on CPU: the rank launcher bench.py uses for `--gpus N` (dist.spawn_ranks: fresh
    processes, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), replica sharding and the final observable
    gather, world_size 2 over gloo"""
    sys.path.insert(0, os.path.join(ROOT, "montecarlo-surfacer_amd"))
    import importlib
    D = importlib.import_module("dist")
    w = tmp_path / "worker.py"
    w.write_text(_GLOO_WORKER)
    assert D.spawn_ranks(str(w), [os.path.join(ROOT, "montecarlo-surfacer_amd")], 2, timeout=240) == 0
    # a failing rank takes the launch down with its exit code instead of leaving the others waiting
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(600)\n")
    import time
    t0 = time.time()
    assert D.spawn_ranks(str(bad), [], 2, timeout=240) == 7 and time.time() - t0 < 60


def test_bench_refuses_rank_count_mismatch():
    """bench.py never runs a rank count other than --gpus: under a launcher that set WORLD_SIZE
    differently it exits non-zero before touching the GPU (round 1 silently ran one rank)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "metric" not in r.stdout
    env.pop("WORLD_SIZE")
    if __import__("smcx_loader").load().device_count() == 0:
        # without WORLD_SIZE the script launches the ranks itself; with no GPU both fail loudly
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--N", "256",
                            "--replicas", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "metric" not in r.stdout
