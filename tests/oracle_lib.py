"""ctypes binding of oracle/libsmc_oracle.so -- the CPU checker.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
RAND_MAX = 2147483647

# wall-strength fixture, SURVEY.md 8a row W (real reference output, glibc 2.35)
W_FIXTURE = np.array([
    962.2264072645321, 57.35316319850277,
    874.39446992695275, 52.11797177356199,
    857.36680597299653, 51.103043912231705,
    1024.1964124687327, 61.046863345428257,
    925.40789594507817, 55.158608910148025,
    913.63518965684239, 54.456900933792724,
    848.90539177252572, 50.598704324515197,
    992.35137245273086, 59.148751047416368,
    844.42493013196849, 50.331648000000015,
], dtype=np.float64)

A0_PLANE = 5.960464477539063e-9   # SMC.h:32
B0_PLANE = 2.44140625e-5          # SMC.h:33


class OrcRng(C.Structure):
    _fields_ = [("s", C.c_uint32 * 31), ("f", C.c_int32), ("r", C.c_int32)]


class OrcSys(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("L", C.c_double), ("Lz", C.c_double),
                ("cutoff", C.c_double), ("a0", C.c_double), ("b0", C.c_double),
                ("Ncx", C.c_int32), ("Ncz", C.c_int32)]


class OrcMoveTrace(C.Structure):
    _fields_ = [("n", C.c_int32), ("accepted", C.c_int32), ("Um", C.c_double),
                ("Fm", C.c_double * 3), ("delta", C.c_double * 3), ("prop", C.c_double * 3),
                ("Un", C.c_double), ("Fn", C.c_double * 3), ("ap", C.c_double), ("u", C.c_double)]


class OrcChainResult(C.Structure):
    _fields_ = [("E0", C.c_double), ("meanE", C.c_double), ("dE", C.c_double),
                ("acceptance_ratio", C.c_double), ("therm_acceptance", C.c_double),
                ("Efinal", C.c_double), ("accepted", C.c_uint64), ("gathers", C.c_uint64),
                ("oob", C.c_uint64), ("cv", C.c_double)]


TRACE_DTYPE = np.dtype([("n", "i4"), ("accepted", "i4"), ("Um", "f8"), ("Fm", "f8", 3),
                        ("delta", "f8", 3), ("prop", "f8", 3), ("Un", "f8"), ("Fn", "f8", 3),
                        ("ap", "f8"), ("u", "f8")], align=True)
assert TRACE_DTYPE.itemsize == C.sizeof(OrcMoveTrace)

_dp = C.POINTER(C.c_double)


class OrcLcaCounts(C.Structure):
    _fields_ = [("n1", C.c_uint64), ("h2", C.c_uint64 * 16), ("h3", C.c_uint64 * 16),
                ("analyses", C.c_uint64), ("overflow", C.c_uint64)]


def _ptr(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libsmc_oracle.so")
    src = os.path.join(ORACLE_DIR, "smc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libsmc_oracle.so"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_srand.argtypes = [C.POINTER(OrcRng), C.c_uint]
        L.orc_rand.argtypes = [C.POINTER(OrcRng)]
        L.orc_rand.restype = C.c_int
        for f in (L.orc_vec_box_muller, L.orc_vec_box_muller_nw):
            f.argtypes = [C.POINTER(OrcRng), C.c_double, C.c_size_t, _dp]
        sp = C.POINTER(OrcSys)
        L.orc_energy_single.argtypes = [sp, _dp, C.c_int]
        L.orc_energy_single.restype = C.c_double
        L.orc_force_single.argtypes = [sp, _dp, C.c_int, _dp]
        L.orc_walls_energy_single.argtypes = [sp, C.c_double, C.c_double, C.c_double, _dp]
        L.orc_walls_energy_single.restype = C.c_double
        L.orc_walls_force.argtypes = [sp, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.orc_energy.argtypes = [sp, _dp]
        L.orc_energy.restype = C.c_double
        L.orc_walls_energy.argtypes = [sp, _dp, _dp]
        L.orc_walls_energy.restype = C.c_double
        L.orc_one_particle_moves.argtypes = [sp, C.POINTER(OrcRng), _dp, _dp, _dp, C.c_double,
                                             C.c_double, C.POINTER(C.c_int), _dp,
                                             C.POINTER(OrcMoveTrace)]
        L.orc_local_density.argtypes = [sp, _dp, C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_chain.argtypes = [sp, C.c_uint, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int,
                                C.c_int, C.c_uint, _dp, C.POINTER(C.c_int32),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64), C.POINTER(OrcChainResult)]
        L.orc_chain.restype = C.c_int
        L.orc_chain_p.argtypes = L.orc_chain.argtypes[:-1] + [_dp, C.POINTER(OrcChainResult)]
        L.orc_chain_p.restype = C.c_int
        L.orc_chain_lca.argtypes = L.orc_chain_p.argtypes[:-1] + [C.c_int, C.c_double,
                                                                 C.POINTER(OrcLcaCounts),
                                                                 C.POINTER(OrcChainResult)]
        L.orc_chain_lca.restype = C.c_int
        L.orc_chain_jt.argtypes = [sp, C.c_uint, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_uint, _dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), _dp, C.POINTER(OrcChainResult)]
        L.orc_chain_jt.restype = C.c_int
        L.orc_cluster_analysis.argtypes = [C.c_int, _dp, C.c_double, C.c_double,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.orc_cluster_analysis.restype = None
        L.orc_cluster_counts.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(OrcLcaCounts)]
        L.orc_cluster_counts.restype = None
        L.orc_pressure.argtypes = [sp, _dp]
        L.orc_pressure.restype = C.c_double
        L.orc_walls_pressure.argtypes = [sp, _dp, _dp]
        L.orc_walls_pressure.restype = C.c_double
        L.orc_fcc_init.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        L.orc_fcc_init.restype = C.c_int
        L.orc_initialize_box_ref.argtypes = [C.c_double, C.c_double, C.c_int, _dp]
        L.orc_initialize_box_ref.restype = C.c_int
        L.orc_initialize_walls.argtypes = [C.c_double] * 4 + [C.c_int, C.c_double, _dp]
        L.orc_nw_energy_single.argtypes = [C.c_int, _dp, C.c_double, C.c_int]
        L.orc_nw_energy_single.restype = C.c_double
        L.orc_nw_force.argtypes = [C.c_int, _dp, C.c_double, C.c_int, _dp]
        L.orc_nw_energy.argtypes = [C.c_int, _dp, C.c_double]
        L.orc_nw_energy.restype = C.c_double
        L.orc_nw_pressure.argtypes = [C.c_int, _dp, C.c_double]
        L.orc_nw_pressure.restype = C.c_double
        L.orc_nw_one_particle_moves.argtypes = [C.c_int, C.POINTER(OrcRng), _dp, _dp, C.c_double,
                                                C.c_double, C.c_double, C.POINTER(C.c_int),
                                                C.POINTER(OrcMoveTrace)]
        L.orc_nw_fcc_init.argtypes = [C.c_int, C.c_double, _dp]
        L.orc_nw_fcc_init.restype = C.c_int
        L.orc_time_sweeps.argtypes = [sp, C.c_uint, _dp, _dp, C.c_double, C.c_double, C.c_int,
                                      C.POINTER(C.c_uint64)]
        L.orc_time_sweeps.restype = C.c_double
        _lib = L
    return _lib


def make_sys(N, M=3, L=33.0, Lz=240.0, cutoff=3.0, a0=A0_PLANE, b0=B0_PLANE, Ncx=33, Ncz=33):
    return OrcSys(N, M, L, Lz, cutoff, a0, b0, Ncx, Ncz)


class Rng:
    def __init__(self, seed):
        self.g = OrcRng()
        lib().orc_srand(C.byref(self.g), seed)

    @classmethod
    def from_state(cls, st):
        """from the engine's exported layout (smcx_rng_export: 31 words oldest first, 0 pending)"""
        assert int(st[31]) == 0
        r = cls(1)
        for i in range(31):
            r.g.s[i] = int(st[i])
        r.g.f, r.g.r = 0, 28   # glibc keeps the front pointer three words ahead of the rear one
        return r

    def rand(self):
        return lib().orc_rand(C.byref(self.g))

    def draws(self, n):
        return np.array([self.rand() for _ in range(n)], dtype=np.int64)

    def box_muller(self, sigma, length, fill=np.nan, nw=False):
        A = np.full(length, fill, dtype=np.float64)
        f = lib().orc_vec_box_muller_nw if nw else lib().orc_vec_box_muller
        f(C.byref(self.g), sigma, length, _ptr(A))
        return A


def fcc(Na, Nz, L=33.0, Lz=240.0):
    N = 4 * Na * Na * Nz
    X = np.zeros(3 * N)
    assert lib().orc_fcc_init(Na, Nz, L, Lz, _ptr(X)) == N
    return X


def box_ref(N, L, Lz):
    X = np.zeros(3 * N)
    placed = lib().orc_initialize_box_ref(L, Lz, N, _ptr(X))
    return X, placed


def walls(M=3, uninit=0.0, x0m=1.6, x0sigma=0.0, ymm=3.0, ymsigma=0.5):
    W = np.zeros(2 * M * M)
    lib().orc_initialize_walls(x0m, x0sigma, ymm, ymsigma, M, uninit, _ptr(W))
    return W


def energy_single(s, R, i):
    return lib().orc_energy_single(C.byref(s), _ptr(R), i)


def force_single(s, R, i):
    F = np.zeros(3)
    lib().orc_force_single(C.byref(s), _ptr(R), i, _ptr(F))
    return F


def walls_energy_single(s, p, W):
    return lib().orc_walls_energy_single(C.byref(s), p[0], p[1], p[2], _ptr(W))


def walls_force(s, p, W, F=None):
    F = np.zeros(3) if F is None else F
    lib().orc_walls_force(C.byref(s), p[0], p[1], p[2], _ptr(W), _ptr(F))
    return F


def total_energy(s, R, W):
    return lib().orc_energy(C.byref(s), _ptr(R)) + lib().orc_walls_energy(C.byref(s), _ptr(R), _ptr(W))


def eval_move(s, R, W, n, prop):
    """Um,Fm at R[n]; Un,Fn with particle n placed at prop (the K1-K4 calls of SMC.c:300-321)."""
    R = np.array(R, dtype=np.float64, copy=True)
    Um = energy_single(s, R, n) + walls_energy_single(s, R[3 * n:3 * n + 3], W)
    Fm = walls_force(s, R[3 * n:3 * n + 3], W, force_single(s, R, n))
    R[3 * n:3 * n + 3] = prop
    Un = energy_single(s, R, n) + walls_energy_single(s, prop, W)
    Fn = walls_force(s, prop, W, force_single(s, R, n))
    return Um, Fm, Un, Fn


def sweep(s, rng, R, W, A, T, E=0.0, trace=False):
    """One oneParticleMoves call. R is updated in place. Returns (accepted, E, trace|None)."""
    N = s.N
    Rn = np.zeros(3 * N)
    j = C.c_int(0)
    Ed = C.c_double(E)
    tr = np.zeros(N, dtype=TRACE_DTYPE) if trace else None
    tp = tr.ctypes.data_as(C.POINTER(OrcMoveTrace)) if trace else None
    lib().orc_one_particle_moves(C.byref(s), C.byref(rng.g), _ptr(R), _ptr(Rn), _ptr(W), A, T,
                                 C.byref(j), C.byref(Ed), tp)
    return j.value, Ed.value, tr


def chain(s, seed, R0, W, T, A, eqsteps, maxsteps, gather_lapse, e0_restart=True, full_hist=False,
          pressure=False, lca_time=0, lca_cutoff=1.7):
    R = np.array(R0, dtype=np.float64, copy=True)
    E = np.zeros(maxsteps + 1)
    jj = np.zeros(max(maxsteps, 1), dtype=np.int32)
    zh = np.zeros(s.Ncz, dtype=np.uint64)
    Nc = s.Ncx * s.Ncx * s.Ncz
    D = np.zeros(Nc, dtype=np.uint64) if full_hist else None
    Mu = np.zeros(Nc, dtype=np.uint64) if full_hist else None
    res = OrcChainResult()
    P = np.zeros(max(maxsteps // gather_lapse, 1)) if pressure else None
    lca = OrcLcaCounts()
    rc = lib().orc_chain_lca(C.byref(s), seed, _ptr(R), _ptr(W), T, A, eqsteps, maxsteps, gather_lapse,
                             1 if e0_restart else 0, _ptr(E), _ptr(jj, C.c_int32),
                             _ptr(zh, C.c_uint64), _ptr(D, C.c_uint64), _ptr(Mu, C.c_uint64),
                             _ptr(P), lca_time, lca_cutoff, C.byref(lca), C.byref(res))
    assert rc == 0
    out = {k: getattr(res, k) for k, _ in OrcChainResult._fields_}
    out["lca"] = dict(n1=int(lca.n1), h2=np.array(lca.h2[:], dtype=np.uint64),
                      h3=np.array(lca.h3[:], dtype=np.uint64), analyses=int(lca.analyses),
                      overflow=int(lca.overflow))
    out.update(R=R, E=E, jj=jj[:maxsteps], zhist=zh, D=D, Mu=Mu,
               P=None if P is None else P[:maxsteps // gather_lapse])
    return out


def chain_jt(s, seed, R0, W, T, A, eqsteps, maxsteps, gather_lapse):
    """the chain with everything sMC keeps: E[], jj[], jt[], D, Mu, P per gather, final reductions"""
    R = np.array(R0, dtype=np.float64, copy=True)
    E = np.zeros(maxsteps + 1)
    jj = np.zeros(max(maxsteps, 1), dtype=np.int32)
    jt = np.zeros(max(eqsteps, 1), dtype=np.int32)
    zh = np.zeros(s.Ncz, dtype=np.uint64)
    Nc = s.Ncx * s.Ncx * s.Ncz
    D = np.zeros(Nc, dtype=np.uint64)
    Mu = np.zeros(Nc, dtype=np.uint64)
    P = np.zeros(max(maxsteps // gather_lapse, 1))
    res = OrcChainResult()
    rc = lib().orc_chain_jt(C.byref(s), seed, _ptr(R), _ptr(W), T, A, eqsteps, maxsteps, gather_lapse, 1,
                            _ptr(E), _ptr(jj, C.c_int32), _ptr(jt, C.c_int32), _ptr(zh, C.c_uint64),
                            _ptr(D, C.c_uint64), _ptr(Mu, C.c_uint64), _ptr(P), C.byref(res))
    assert rc == 0
    out = {k: getattr(res, k) for k, _ in OrcChainResult._fields_}
    out.update(R=R, E=E, jj=jj[:maxsteps], jt=jt[:eqsteps], zhist=zh, D=D, Mu=Mu,
               P=P[:maxsteps // gather_lapse])
    return out


def cluster_analysis(N, R, L, cutoff=1.7):
    """orc_cluster_analysis: (LCA[N(N-1)/2][3], overflow)"""
    R = np.ascontiguousarray(R, dtype=np.float64)
    LCA = np.zeros((N * (N - 1) // 2, 3), dtype=np.int32)
    ov = C.c_uint64(0)
    lib().orc_cluster_analysis(N, _ptr(R), L, cutoff, _ptr(LCA, C.c_int32), C.byref(ov))
    return LCA, ov.value


def cluster_counts(N, LCA):
    c = OrcLcaCounts()
    LCA = np.ascontiguousarray(LCA, dtype=np.int32)
    lib().orc_cluster_counts(N, _ptr(LCA, C.c_int32), C.byref(c))
    return int(c.n1), np.array(c.h2[:], dtype=np.uint64), np.array(c.h3[:], dtype=np.uint64)


def pressure(s, R, W):
    return lib().orc_pressure(C.byref(s), _ptr(R)) + lib().orc_walls_pressure(C.byref(s), _ptr(R), _ptr(W))


def time_sweeps(s, seed, R, W, T, A, sweeps):
    acc = C.c_uint64(0)
    t = lib().orc_time_sweeps(C.byref(s), seed, _ptr(R), _ptr(W), T, A, sweeps, C.byref(acc))
    return t, acc.value


def fft_acf(H, k_max):
    """numpy restatement of the reference's fft_acf (SMC.c:1051-1089; FFTW is absent from the
    image, so this row is pinned only against a direct O(n^2) evaluation of the same sums):
    r2c transform of H - mean(H), |.|^2 of its first lfft = n/2 + n%2 bins, backward complex
    transform of THAT length (FFTW_BACKWARD is unnormalised), acf[i] = Re C[i] / Re C[0]."""
    H = np.asarray(H, dtype=np.float64)
    n = len(H)
    if n < k_max * 2 + 1:
        k_max = n // 2 - 2                                     # SMC.c:1054-1057
    lfft = n // 2 + n % 2                                      # SMC.c:1063
    s = 0.0
    for v in H:                                                # mean() adds in index order
        s += v
    Z = H - s / n
    F = np.fft.rfft(Z)[:lfft]
    T = (F * np.conj(F)).real + 0.0j
    Cc = np.fft.ifft(T) * lfft
    return (Cc.real[:k_max] / Cc.real[0]).copy()


def fft_acf_direct(H, k_max):
    """the same quantity from the defining sums, no FFT (O(n^2); small n only)"""
    H = np.asarray(H, dtype=np.float64)
    n = len(H)
    if n < k_max * 2 + 1:
        k_max = n // 2 - 2
    lfft = n // 2 + n % 2
    Z = H - H.sum() / n
    j = np.arange(n)
    F = np.array([(Z * np.exp(-2j * np.pi * k * j / n)).sum() for k in range(lfft)])
    T = np.abs(F) ** 2
    k = np.arange(lfft)
    Cc = np.array([(T * np.exp(2j * np.pi * k * i / lfft)).sum() for i in range(k_max)])
    return Cc.real / Cc.real[0]
