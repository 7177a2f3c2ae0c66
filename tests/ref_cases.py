"""The cases that pin the oracle on the REAL reference (SMC.c / SMC_noMPI_noWall.c compiled from
/root/reference into oracle/_ref by oracle/build_ref.sh).  TEST INFRASTRUCTURE.

`compute(backend, cases)` runs one list of cases through a backend and returns plain JSON data:
floats as C99 hex strings, large arrays as sha256 of their bytes (parity on this path is
bit-for-bit, so a digest loses nothing).  Two backends exist:
  * RefBackend    -- the real reference functions (tests/ref_lib.py); build container only;
  * OracleBackend -- oracle/smc_oracle.c (tests/oracle_lib.py), available everywhere.
tests/golden/make_ref_golden.py stores compute(RefBackend, GOLDEN_CASES) in
tests/golden/ref_smc.json; tests/test_ref_pin.py checks compute(OracleBackend, GOLDEN_CASES)
against that file everywhere, and compute(OracleBackend, LIVE_CASES) against
compute(RefBackend, LIVE_CASES) wherever oracle/_ref is present.

Inputs are built by deterministic numpy code in this file only (lattices, seeds), never by
either backend, so both sides see identical bytes.
"""
import hashlib

import numpy as np

L_BOX, LZ_BOX, T_REF, A_REF = 33.0, 240.0, 1.1, 1.1   # main.c:41-51, SURVEY 8d


def hx(v):
    return float(v).hex()


def hexes(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def digest(a):
    a = np.ascontiguousarray(a)
    return "%s:%s" % (a.dtype.str, hashlib.sha256(a.tobytes()).hexdigest())


def fcc(Na, Nz, L=L_BOX, Lz=LZ_BOX):
    """build-defined fcc(Na,Nz) start of SURVEY 8d: the cell order and +a/4 of SMC.c:432-458,
    then shiftSystem3D(X, L, 0.95 Lz) (SMC.c:461), written here in numpy with the same
    expression per coordinate (a*i, + a/2, + a/4, x - L*rint(x/L))."""
    a = L / Na
    X = np.zeros((Na * Na * Nz, 4, 3))
    i, j, k = np.meshgrid(np.arange(Na), np.arange(Na), np.arange(Nz), indexing="ij")
    base = np.stack([a * i.ravel(), a * j.ravel(), a * k.ravel()], axis=1)
    X[:, 0] = base
    X[:, 1] = base + np.array([a / 2, a / 2, 0.0])
    X[:, 2] = base + np.array([a / 2, 0.0, a / 2])
    X[:, 3] = base + np.array([0.0, a / 2, a / 2])
    X = X.reshape(-1, 3) + a / 4
    Lz95 = Lz - Lz / 20.0
    X[:, 0] = X[:, 0] - L * np.rint(X[:, 0] / L)
    X[:, 1] = X[:, 1] - L * np.rint(X[:, 1] / L)
    X[:, 2] = X[:, 2] - Lz95 * np.rint(X[:, 2] / Lz95)
    return X.ravel().copy()


def jittered(X, seed, amp, L=L_BOX):
    """a lattice with reproducible noise (numpy's MT19937), x,y wrapped as SMC.c:315-316"""
    rs = np.random.RandomState(seed)
    Y = X.reshape(-1, 3) + amp * rs.standard_normal((X.size // 3, 3))
    Y[:, 0] = Y[:, 0] - L * np.rint(Y[:, 0] / L)
    Y[:, 1] = Y[:, 1] - L * np.rint(Y[:, 1] / L)
    return Y.ravel().copy()


def start_state(spec):
    """spec: ("fcc", Na, Nz[, L, Lz]) | ("jit", Na, Nz, seed, amp[, L, Lz]) | ("gas", N, seed[, L, Lz, zspan])
    | ("film", Na, Nz, zcentre, seed, amp)"""
    kind = spec[0]
    if kind == "fcc":
        return fcc(*spec[1:])
    if kind == "jit":
        Na, Nz, seed, amp = spec[1:5]
        return jittered(fcc(Na, Nz, *spec[5:]), seed, amp, *(spec[5:6]))
    if kind == "gas":
        N, seed = spec[1:3]
        L = spec[3] if len(spec) > 3 else L_BOX
        Lz = spec[4] if len(spec) > 4 else LZ_BOX
        zs = spec[5] if len(spec) > 5 else 0.9
        rs = np.random.RandomState(seed)
        Y = np.empty((N, 3))
        Y[:, 0] = (rs.random_sample(N) - 0.5) * L
        Y[:, 1] = (rs.random_sample(N) - 0.5) * L
        Y[:, 2] = (rs.random_sample(N) - 0.5) * Lz * zs
        return Y.ravel().copy()
    if kind == "film":
        Na, Nz, zc, seed, amp = spec[1:6]
        X = jittered(fcc(Na, Nz), seed, amp).reshape(-1, 3)
        X[:, 2] += zc - X[:, 2].mean()
        return X.ravel().copy()
    raise ValueError(kind)


# ------------------------------------------------------------------------------------------
# case lists.  Every case is a dict of plain values; "N" selects the reference build.
# ------------------------------------------------------------------------------------------
def _wall_points(L, Lz):
    pts = []
    for z in (Lz / 2 - 1e-3, -(Lz / 2 - 1e-3), Lz / 2 - 0.9, -(Lz / 2 - 1.2), Lz / 2 - 2.9, -(Lz / 2 - 3.05),
              Lz / 2, -Lz / 2, Lz / 2 + 0.5, -(Lz / 2 + 2.0), 0.0, 17.0, Lz / 2 - 1e-5):
        for (x, y) in ((0.0, 0.0), (L / 3, L / 3), (-L / 2, L / 2 - 0.01), (5.4, -11.2), (L / 2, -L / 2)):
            pts.append((x, y, z))
    return pts


GOLDEN_CASES = [
    {"kind": "walls", "N": 108},
    {"kind": "box", "N": 32, "L": 33.0, "Lz": 200.0},
    {"kind": "box", "N": 108, "L": 33.0, "Lz": 200.0},
    {"kind": "box", "N": 256, "L": 33.0, "Lz": 240.0},
    {"kind": "box", "N": 500, "L": 33.0, "Lz": 240.0},
    {"kind": "box", "N": 1024, "L": 33.0, "Lz": 240.0},     # 16 particles left unplaced
    {"kind": "box", "N": 4000, "L": 33.0, "Lz": 240.0},
    {"kind": "box", "N": 4096, "L": 33.0, "Lz": 240.0},     # 96 unplaced
    {"kind": "box", "N": 16384, "L": 33.0, "Lz": 240.0},
    {"kind": "wall_points", "N": 108, "L": 33.0, "Lz": 240.0},
    {"kind": "wall_points", "N": 256, "L": 20.0, "Lz": 31.0},
    # K1-K4 + K5 + pressure on states with pairs inside the cutoff
    {"kind": "single", "N": 256, "start": ("jit", 4, 4, 11, 0.4), "particles": [0, 1, 17, 100, 128, 255]},
    {"kind": "single", "N": 256, "start": ("gas", 256, 5, 9.0, 12.0, 1.05), "L": 9.0, "Lz": 12.0,
     "particles": list(range(0, 256, 9))},                   # dense gas: clamp branch, overlaps, images
    {"kind": "single", "N": 1024, "start": ("fcc", 8, 4), "particles": [0, 3, 512, 1023]},
    {"kind": "single", "N": 1024, "start": ("jit", 8, 4, 12, 0.3), "particles": [0, 1, 2, 3, 500, 777, 1023]},
    {"kind": "single", "N": 1024, "start": ("film", 16, 1, 118.9, 13, 0.1), "particles": [0, 5, 600, 1023]},
    {"kind": "single", "N": 4096, "start": ("jit", 8, 16, 14, 0.3), "particles": [0, 63, 64, 2047, 4095]},
    {"kind": "single", "N": 4096, "start": ("jit", 16, 4, 15, 0.05), "particles": [0, 1000, 4095]},
    {"kind": "single", "N": 16384, "start": ("jit", 16, 16, 16, 0.05), "particles": [0, 8191, 16383]},
    # chains through the real oneParticleMoves + localDensityAndMobility + pressure, sMC's loop
    {"kind": "chain", "N": 108, "start": ("refbox",), "L": 33.0, "Lz": 200.0, "seed": 12345,
     "eq": 0, "steps": 20, "lapse": 1},                      # the chain SURVEY 8c recorded
    {"kind": "chain", "N": 32, "start": ("refbox",), "L": 6.0, "Lz": 12.0, "seed": 7, "eq": 3, "steps": 12, "lapse": 2},
    {"kind": "chain", "N": 256, "start": ("fcc", 4, 4), "seed": 12345, "eq": 0, "steps": 20, "lapse": 1},
    {"kind": "chain", "N": 256, "start": ("fcc", 4, 4), "seed": 12352, "eq": 2, "steps": 6, "lapse": 2},   # smoke()
    {"kind": "chain", "N": 256, "start": ("jit", 4, 4, 21, 0.4), "seed": 99, "eq": 5, "steps": 10, "lapse": 3},
    {"kind": "chain", "N": 256, "start": ("gas", 256, 6, 9.0, 12.0, 0.98), "L": 9.0, "Lz": 12.0, "seed": 4,
     "eq": 2, "steps": 8, "lapse": 1},                       # dense: rejections, wall clamp, uint8 cells
    {"kind": "chain", "N": 500, "start": ("refbox",), "seed": 1, "eq": 1, "steps": 4, "lapse": 2},
    {"kind": "chain", "N": 1024, "start": ("fcc", 8, 4), "seed": 12345, "eq": 0, "steps": 10, "lapse": 1},
    {"kind": "chain", "N": 1024, "start": ("fcc", 8, 4), "seed": 12348, "eq": 0, "steps": 4, "lapse": 2},  # smoke()
    {"kind": "chain", "N": 1024, "start": ("film", 16, 1, 118.9, 13, 0.1), "seed": 3, "eq": 1, "steps": 3, "lapse": 1},
    {"kind": "chain", "N": 4000, "start": ("refbox",), "seed": 12345, "eq": 0, "steps": 2, "lapse": 1},
    {"kind": "chain", "N": 4096, "start": ("fcc", 8, 16), "seed": 12345, "eq": 0, "steps": 3, "lapse": 1},
    {"kind": "chain", "N": 4096, "start": ("fcc", 8, 16), "seed": 12346, "eq": 2, "steps": 3, "lapse": 1},
    {"kind": "chain", "N": 4096, "start": ("fcc", 16, 4), "seed": 12345, "eq": 0, "steps": 1, "lapse": 1},
    {"kind": "chain", "N": 16384, "start": ("fcc", 16, 16), "seed": 12345, "eq": 0, "steps": 1, "lapse": 1},
    # (gases with ~1.5 bonds per particle: on a crystal the reference's common_nn[8] overflows its stack
    # frame -- entries shared between rows accumulate num2, SMC.c:985, 1004-1016 -- and the process dies)
    {"kind": "cluster", "N": 108, "start": ("gas", 108, 8, 11.0, 12.0, 1.0), "L": 11.0},
    {"kind": "cluster", "N": 256, "start": ("gas", 256, 31, 15.0, 16.0, 1.0), "L": 15.0},
    {"kind": "nw", "N": 256, "rho": 0.1, "T": 0.4, "A": 4e-8, "seed": 12345, "steps": 10},   # BASELINE config 1
    {"kind": "nw", "N": 256, "rho": 0.1, "T": 0.4, "A": 0.02, "seed": 12345, "steps": 10},   # moves that matter
    {"kind": "nw", "N": 108, "rho": 0.8, "T": 1.0, "A": 0.002, "seed": 5, "steps": 8},
    {"kind": "nw", "N": 32, "rho": 0.5, "T": 0.7, "A": 0.01, "seed": 77, "steps": 15},
]


def live_cases(seed0):
    """a second, seed-dependent set for the live comparison (reference libraries present)"""
    rs = np.random.RandomState(seed0)
    cs = []
    for N, (na, nz) in ((256, (4, 4)), (1024, (8, 4)), (4096, (8, 16))):
        s = int(rs.randint(1, 2 ** 31 - 1))
        cs.append({"kind": "single", "N": N, "start": ("jit", na, nz, s % 1000, 0.35),
                   "particles": [int(v) for v in rs.randint(0, N, size=6)]})
        cs.append({"kind": "chain", "N": N, "start": ("jit", na, nz, s % 977, 0.2), "seed": s,
                   "eq": int(rs.randint(0, 3)), "steps": 12 if N == 256 else (5 if N == 1024 else 2),
                   "lapse": int(rs.randint(1, 3))})
    cs.append({"kind": "chain", "N": 256, "start": ("gas", 256, int(rs.randint(1000)), 9.0, 12.0, 1.0), "L": 9.0,
               "Lz": 12.0, "seed": int(rs.randint(1, 2 ** 31 - 1)), "eq": 1, "steps": 6, "lapse": 1})
    cs.append({"kind": "chain", "N": 4096, "start": ("film", 16, 4, -110.0, int(rs.randint(1000)), 0.08),
               "seed": int(rs.randint(1, 2 ** 31 - 1)), "eq": 0, "steps": 1, "lapse": 1})
    cs.append({"kind": "chain", "N": 1024, "start": ("fcc", 8, 4), "seed": 4294967295, "eq": 0, "steps": 2, "lapse": 1})
    cs.append({"kind": "nw", "N": 256, "rho": 0.6, "T": 0.9, "A": 0.004, "seed": int(rs.randint(1, 2 ** 31 - 1)),
               "steps": 6})
    cs.append({"kind": "cluster", "N": 256, "start": ("gas", 256, int(rs.randint(1000)), 15.0, 16.0, 1.0), "L": 15.0})
    return cs


# ------------------------------------------------------------------------------------------
def compute(backend, cases, progress=None):
    out = []
    for c in cases:
        if progress:
            progress(c)
        out.append(getattr(backend, "case_" + c["kind"])(c))
    return out


class _Common:
    """what both backends share: the bookkeeping around a case, not the arithmetic"""

    def _chain_record(self, c, r):
        return {"E0": hx(r["E"][0]), "E": hexes(r["E"]), "jj": [int(v) for v in r["jj"]],
                "jt": [int(v) for v in r["jt"]], "R": digest(r["R"]),
                "R_head": hexes(r["R"][:6]), "zhist": [int(v) for v in r["zhist"]],
                "D": digest(r["D"].astype(np.uint64)), "Mu": digest(r["Mu"].astype(np.uint64)),
                "oob": int(r["oob"]), "P": hexes(r["P"]), "meanE": hx(r["meanE"]), "dE": hx(r["dE"]),
                "acceptance_ratio": hx(r["acceptance_ratio"]), "cv": hx(r["cv"])}


class RefBackend(_Common):
    def __init__(self):
        import ref_lib
        self.RL = ref_lib
        self._smc, self._nw = {}, {}

    def smc(self, n):
        if n not in self._smc:
            self._smc[n] = self.RL.RefSMC(n)
        return self._smc[n]

    def nw(self, n):
        if n not in self._nw:
            self._nw[n] = self.RL.RefNW(n)
        return self._nw[n]

    def W(self):
        return self.smc(108).initialize_walls()

    def case_walls(self, c):
        return {"W": hexes(self.W())}

    def case_box(self, c):
        X = self.smc(c["N"]).initialize_box(c["L"], c["Lz"])
        return {"X": digest(X), "head": hexes(X[:12]), "tail": hexes(X[-6:])}

    def case_wall_points(self, c):
        r, W = self.smc(c["N"]), self.W()
        res = []
        for k, (x, y, z) in enumerate(_wall_points(c["L"], c["Lz"])):
            e, F = r.wall_point(x, y, z, W, c["L"], c["Lz"], Fin=(0.25 * k, -1.0, 3.0))
            res.append(hexes([e, F[0], F[1], F[2]]))
        return {"points": res}

    def _start(self, c):
        L, Lz = c.get("L", L_BOX), c.get("Lz", LZ_BOX)
        if c["start"][0] == "refbox":
            return self.smc(c["N"]).initialize_box(L, Lz), L, Lz
        X = start_state(c["start"])
        assert X.size == 3 * c["N"]
        return X, L, Lz

    def case_single(self, c):
        r, W = self.smc(c["N"]), self.W()
        X, L, Lz = self._start(c)
        vals = []
        for i in c["particles"]:
            e, ew, F, Ft = r.single(X, W, L, Lz, i)
            vals.append(hexes([e, ew, *F, *Ft]))
        return {"single": vals, "energy": hx(r.energy(X, L)), "walls_energy": hx(r.walls_energy(X, W, L, Lz)),
                "pressure": hx(r.pressure(X, L, Lz)), "walls_pressure": hx(r.walls_pressure(X, W, L, Lz))}

    def case_chain(self, c):
        r, W = self.smc(c["N"]), self.W()
        X, L, Lz = self._start(c)
        res = r.chain(c["seed"], X, W, L, Lz, c.get("T", T_REF), c.get("A", A_REF), c["eq"], c["steps"], c["lapse"])
        return self._chain_record(c, res)

    def case_cluster(self, c):
        r = self.smc(c["N"])
        X = start_state(c["start"])
        LCA = r.cluster_analysis(X, c["L"])
        return {"LCA": digest(LCA), "bonded": int((LCA[0::3] != 0).sum()),
                "num2": int(LCA[1::3].sum()), "num3": int(LCA[2::3].sum())}

    def case_nw(self, c):
        r = self.nw(c["N"])
        L = float(np.cbrt(c["N"] / c["rho"]))
        X = r.initialize_box(L)
        rec = {"L": hx(L), "X": digest(X), "E0": hx(r.energy(X, L)), "P0": hx(r.pressure(X, L)),
               "single": [hexes([*(lambda e, F: (e, *F))(*r.single(X, L, i))]) for i in (0, 1, c["N"] - 1)]}
        jj, Es, Rs = r.sweeps(c["seed"], X, L, c["A"], c["T"], c["steps"], keep_positions=True)
        rec.update(jj=[int(v) for v in jj], Es=hexes(Es), Rs=[digest(x) for x in Rs], P_end=hx(r.pressure(X, L)))
        return rec


class OracleBackend(_Common):
    def __init__(self):
        import oracle_lib
        self.O = oracle_lib
        oracle_lib.lib()

    def W(self):
        return self.O.walls(3, 0.0)

    def case_walls(self, c):
        return {"W": hexes(self.W())}

    def case_box(self, c):
        X, _ = self.O.box_ref(c["N"], c["L"], c["Lz"])
        return {"X": digest(X), "head": hexes(X[:12]), "tail": hexes(X[-6:])}

    def case_wall_points(self, c):
        O, W = self.O, self.W()
        s = O.make_sys(c["N"], L=c["L"], Lz=c["Lz"])
        res = []
        for k, p in enumerate(_wall_points(c["L"], c["Lz"])):
            e = O.walls_energy_single(s, p, W)
            F = O.walls_force(s, p, W, np.array([0.25 * k, -1.0, 3.0]))
            res.append(hexes([e, F[0], F[1], F[2]]))
        return {"points": res}

    def _start(self, c):
        L, Lz = c.get("L", L_BOX), c.get("Lz", LZ_BOX)
        if c["start"][0] == "refbox":
            return self.O.box_ref(c["N"], L, Lz)[0], L, Lz
        X = start_state(c["start"])
        assert X.size == 3 * c["N"]
        return X, L, Lz

    def case_single(self, c):
        import ctypes as C
        O, W = self.O, self.W()
        X, L, Lz = self._start(c)
        s = O.make_sys(c["N"], L=L, Lz=Lz)
        vals = []
        for i in c["particles"]:
            e = O.energy_single(s, X, i)
            p = X[3 * i:3 * i + 3]
            ew = O.walls_energy_single(s, p, W)
            F = O.force_single(s, X, i)
            Ft = O.walls_force(s, p, W, F.copy())
            vals.append(hexes([e, ew, *F, *Ft]))
        lib = O.lib()
        return {"single": vals, "energy": hx(lib.orc_energy(C.byref(s), O._ptr(X))),
                "walls_energy": hx(lib.orc_walls_energy(C.byref(s), O._ptr(X), O._ptr(W))),
                "pressure": hx(lib.orc_pressure(C.byref(s), O._ptr(X))),
                "walls_pressure": hx(lib.orc_walls_pressure(C.byref(s), O._ptr(X), O._ptr(W)))}

    def case_chain(self, c):
        O, W = self.O, self.W()
        X, L, Lz = self._start(c)
        s = O.make_sys(c["N"], L=L, Lz=Lz)
        T, A = c.get("T", T_REF), c.get("A", A_REF)
        return self._chain_record(c, O.chain_jt(s, c["seed"], X, W, T, A, c["eq"], c["steps"], c["lapse"]))

    def case_cluster(self, c):
        X = start_state(c["start"])
        LCA, _ = self.O.cluster_analysis(c["N"], X, c["L"], 1.7)
        LCA = LCA.ravel()
        return {"LCA": digest(LCA), "bonded": int((LCA[0::3] != 0).sum()),
                "num2": int(LCA[1::3].sum()), "num3": int(LCA[2::3].sum())}

    def case_nw(self, c):
        import ctypes as C
        O = self.O
        lib = O.lib()
        N = c["N"]
        L = float(np.cbrt(N / c["rho"]))
        X = np.zeros(3 * N)
        lib.orc_nw_fcc_init(N, L, O._ptr(X))

        def single(i):
            F = np.zeros(3)
            e = lib.orc_nw_energy_single(N, O._ptr(X), L, i)
            lib.orc_nw_force(N, O._ptr(X), L, i, O._ptr(F))
            return hexes([e, *F])
        rec = {"L": hx(L), "X": digest(X), "E0": hx(lib.orc_nw_energy(N, O._ptr(X), L)),
               "P0": hx(lib.orc_nw_pressure(N, O._ptr(X), L)), "single": [single(i) for i in (0, 1, N - 1)]}
        rng = O.Rng(c["seed"])
        Rn = np.zeros(3 * N)
        jj, Es, Rs = [], [], []
        for _ in range(c["steps"]):
            j = C.c_int(0)
            lib.orc_nw_one_particle_moves(N, C.byref(rng.g), O._ptr(X), O._ptr(Rn), L, c["A"], c["T"],
                                          C.byref(j), None)
            jj.append(j.value)
            Es.append(lib.orc_nw_energy(N, O._ptr(X), L))
            Rs.append(digest(X))
        rec.update(jj=jj, Es=hexes(Es), Rs=Rs, P_end=hx(lib.orc_nw_pressure(N, O._ptr(X), L)))
        return rec
