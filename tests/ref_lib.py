"""ctypes binding of oracle/_ref/libref_smc_N<n>.so and libref_nw_N<n>.so: the REAL reference
functions (SMC.c / SMC_noMPI_noWall.c line ranges compiled from /root/reference where they lie
by oracle/build_ref.sh, one library per compile-time N).

TEST INFRASTRUCTURE: used by tests/golden/make_ref_golden.py (fixture generator, build container
only) and by tests/test_ref_pin.py, which checks the oracle against these libraries live when
they are present and against the committed fixtures always.  The product never touches it.
All libraries share libc's one hidden rand() state, exactly like the reference.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_ulp = C.POINTER(C.c_ulong)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def available(n, nw=False):
    return os.path.exists(os.path.join(REF_DIR, "libref_%s_N%d.so" % ("nw" if nw else "smc", n)))


class RefSMC:
    """The walls variant at one compile-time N."""

    def __init__(self, n):
        L = C.CDLL(os.path.join(REF_DIR, "libref_smc_N%d.so" % n))
        for f in ("refw_N", "refw_M", "refw_ncx", "refw_ncz", "refw_lca_time", "refw_rand"):
            getattr(L, f).restype = C.c_int
        for f in ("refw_cutoff", "refw_a0", "refw_b0", "refw_lca_cutoff"):
            getattr(L, f).restype = C.c_double
        L.refw_srand.argtypes = [C.c_uint]
        L.refw_single.argtypes = [_dp, _dp, C.c_double, C.c_double, C.c_int, _dp]
        L.refw_wall_point.argtypes = [C.c_double] * 3 + [_dp, C.c_double, C.c_double, _dp, _dp]
        L.refw_energy.argtypes = [_dp, C.c_double]
        L.refw_energy.restype = C.c_double
        L.refw_walls_energy.argtypes = [_dp, _dp, C.c_double, C.c_double]
        L.refw_walls_energy.restype = C.c_double
        L.refw_pressure.argtypes = [_dp, C.c_double, C.c_double]
        L.refw_pressure.restype = C.c_double
        L.refw_walls_pressure.argtypes = [_dp, _dp, C.c_double, C.c_double]
        L.refw_walls_pressure.restype = C.c_double
        L.refw_sweeps.argtypes = [C.c_long, _dp, _dp, _dp] + [C.c_double] * 4 + [C.c_int, _ip, _dp]
        L.refw_local_density.argtypes = [_dp, C.c_double, C.c_double, _ulp, _ip, _ulp]
        L.refw_hist_len.restype = C.c_long
        L.refw_chain.argtypes = [C.c_uint, _dp, _dp] + [C.c_double] * 4 + [C.c_int] * 3 + \
            [_dp, _ip, _ip, _ulp, _ulp, _dp]
        L.refw_chain.restype = C.c_int
        L.refw_results.argtypes = [_dp, _ip, C.c_int, C.c_double, _dp]
        L.refw_initialize_box.argtypes = [C.c_double, C.c_double, _dp]
        L.refw_initialize_walls.argtypes = [C.c_double] * 4 + [_dp]
        L.refw_initialize_walls.restype = C.c_int
        L.refw_cluster_analysis.argtypes = [_dp, C.c_double, _ip]
        L.refw_simple_acf.argtypes = [_dp, C.c_size_t, C.c_int, _dp]
        self.L = L
        self.N = L.refw_N()
        assert self.N == n
        self.M = L.refw_M()
        self.ncx, self.ncz = L.refw_ncx(), L.refw_ncz()
        self.cutoff, self.a0, self.b0 = L.refw_cutoff(), L.refw_a0(), L.refw_b0()
        self.hist_len = L.refw_hist_len()

    def single(self, r, W, L, Lz, i):
        """(energySingle, wallsEnergySingle, F of forceSingle, F after wallsForce) of particle i."""
        out = np.zeros(8)
        self.L.refw_single(_p(r), _p(W), L, Lz, int(i), _p(out))
        return out[0], out[1], out[2:5].copy(), out[5:8].copy()

    def wall_point(self, x, y, z, W, L, Lz, Fin=(0.0, 0.0, 0.0)):
        out = np.zeros(4)
        fin = np.array(Fin, dtype=np.float64)
        self.L.refw_wall_point(x, y, z, _p(W), L, Lz, _p(fin), _p(out))
        return out[0], out[1:4].copy()

    def energy(self, r, L):
        return self.L.refw_energy(_p(r), L)

    def walls_energy(self, r, W, L, Lz):
        return self.L.refw_walls_energy(_p(r), _p(W), L, Lz)

    def pressure(self, r, L, Lz):
        return self.L.refw_pressure(_p(r), L, Lz)

    def walls_pressure(self, r, W, L, Lz):
        return self.L.refw_walls_pressure(_p(r), _p(W), L, Lz)

    def sweeps(self, seed, R, W, L, Lz, A, T, nsweeps, E0):
        """nsweeps real oneParticleMoves calls from srand(seed) (None: keep the rand() state);
        R is updated in place; returns (E[nsweeps+1], jj[nsweeps])."""
        Rn = np.zeros_like(R)
        E = np.zeros(nsweeps + 1)
        E[0] = E0
        jj = np.zeros(nsweeps, dtype=np.int32)
        self.L.refw_sweeps(-1 if seed is None else int(seed), _p(R), _p(Rn), _p(W), L, Lz, A, T,
                           nsweeps, _p(jj, C.c_int), _p(E))
        return E, jj

    def local_density(self, r, L, Lz, D, Rbin, Mu):
        self.L.refw_local_density(_p(r), L, Lz, _p(D, C.c_ulong), _p(Rbin, C.c_int), _p(Mu, C.c_ulong))

    def chain(self, seed, R0, W, L, Lz, T, A, eqsteps, maxsteps, gather_lapse, pressure=True):
        R = np.array(R0, dtype=np.float64)
        E = np.zeros(max(maxsteps, eqsteps) + 1)
        jj = np.zeros(max(maxsteps, 1), dtype=np.int32)
        jt = np.zeros(max(eqsteps, 1), dtype=np.int32)
        D = np.zeros(self.hist_len, dtype=np.uint64)
        Mu = np.zeros(self.hist_len, dtype=np.uint64)
        P = np.zeros(maxsteps // gather_lapse + 1)
        rc = self.L.refw_chain(int(seed), _p(R), _p(W), L, Lz, T, A, eqsteps, maxsteps, gather_lapse,
                               _p(E), _p(jj, C.c_int), _p(jt, C.c_int), _p(D, C.c_ulong),
                               _p(Mu, C.c_ulong), _p(P) if pressure else None)
        assert rc == 0
        E = E[:maxsteps + 1].copy()
        res = np.zeros(4)
        Ec = E.copy()
        self.L.refw_results(_p(Ec), _p(jj, C.c_int), maxsteps, T, _p(res))
        ncell = self.ncx * self.ncx * self.ncz
        return {"R": R, "E": E, "jj": jj[:maxsteps].copy(), "jt": jt[:eqsteps].copy(),
                "D": D[:ncell].copy(), "Mu": Mu[:ncell].copy(), "oob": int(D[ncell:].sum()),
                "P": P[1:].copy(), "meanE": res[0], "dE": res[1], "acceptance_ratio": res[2],
                "cv": res[3],
                "zhist": D[:ncell].reshape(self.ncx, self.ncx, self.ncz).sum(axis=(0, 1))}

    def initialize_box(self, L, Lz):
        X = np.zeros(3 * self.N)
        self.L.refw_initialize_box(L, Lz, _p(X))
        return X

    def initialize_walls(self, x0m=1.6, x0sigma=0.0, ymm=3.0, ymsigma=0.5):
        W = np.zeros(2 * self.M * self.M)
        assert self.L.refw_initialize_walls(x0m, x0sigma, ymm, ymsigma, _p(W)) == 0
        return W

    def cluster_analysis(self, r, L):
        n = self.N
        LCA = np.zeros(3 * (n * (n - 1) // 2), dtype=np.int32)
        self.L.refw_cluster_analysis(_p(r), L, _p(LCA, C.c_int))
        return LCA

    def simple_acf(self, H, kmax):
        acf = np.zeros(kmax)
        self.L.refw_simple_acf(_p(H), len(H), kmax, _p(acf))
        return acf


class RefNW:
    """SMC_noMPI_noWall.c at one compile-time N."""

    def __init__(self, n):
        L = C.CDLL(os.path.join(REF_DIR, "libref_nw_N%d.so" % n))
        L.refnw_N.restype = C.c_int
        L.refnw_srand.argtypes = [C.c_uint]
        L.refnw_initialize_box.argtypes = [C.c_double, _dp]
        L.refnw_single.argtypes = [_dp, C.c_double, C.c_int, _dp]
        L.refnw_energy.argtypes = [_dp, C.c_double]
        L.refnw_energy.restype = C.c_double
        L.refnw_pressure.argtypes = [_dp, C.c_double]
        L.refnw_pressure.restype = C.c_double
        L.refnw_sweeps.argtypes = [C.c_long, _dp, _dp] + [C.c_double] * 3 + [C.c_int, _ip, _dp, _dp]
        L.refnw_vec_box_muller.argtypes = [C.c_double, C.c_size_t, _dp]
        self.L = L
        self.N = L.refnw_N()
        assert self.N == n

    def initialize_box(self, L):
        X = np.zeros(3 * self.N)
        self.L.refnw_initialize_box(L, _p(X))
        return X

    def single(self, r, L, i):
        out = np.zeros(4)
        self.L.refnw_single(_p(r), L, int(i), _p(out))
        return out[0], out[1:4].copy()

    def energy(self, r, L):
        return self.L.refnw_energy(_p(r), L)

    def pressure(self, r, L):
        return self.L.refnw_pressure(_p(r), L)

    def sweeps(self, seed, R, L, A, T, nsweeps, keep_positions=False):
        Rn = np.zeros_like(R)
        jj = np.zeros(nsweeps, dtype=np.int32)
        Es = np.zeros(nsweeps)
        Rs = np.zeros((nsweeps, 3 * self.N)) if keep_positions else None
        self.L.refnw_sweeps(-1 if seed is None else int(seed), _p(R), _p(Rn), L, A, T, nsweeps,
                            _p(jj, C.c_int), _p(Es), _p(Rs))
        return jj, Es, Rs

    def vec_box_muller(self, seed, sigma, length):
        A = np.zeros(length)
        self.L.refnw_srand(int(seed))
        self.L.refnw_vec_box_muller(sigma, length, _p(A))
        return A
