"""The z-ordered kernels where their RARE paths run, against the ORACLE (which tests/test_ref_pin.py pins bit for
bit on the real SMC.c), not against another kernel: eight replicas (four at N > 8192), one to four sweeps each (0.2 s of
oracle per N = 4096 sweep, one chain per host core).  Reference semantics at stake: the acceptance test SMC.c:326-335 and the
all-neighbour sums SMC.c:557-618 -- a dropped candidate shifts E by >= 4|V(rc)| = 5e-3 and sooner or later flips a decision.
Required per case: accepted count of every sweep equal, z histogram equal, energy series and final positions within the
schedule of tests/tolerances.py.
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import os

import numpy as np
import pytest

import tolerances as TOL

pytestmark = pytest.mark.gpu

T = A = 1.1
S_KERNEL = {"ma64": 5, "mb64": 6}   # SMCX_KERNEL_MA, SMCX_KERNEL_MB (include/smcx.h)


def _sys(O, p):
    return O.make_sys(p.N, M=p.M, L=p.L, Lz=p.Lz, cutoff=p.cutoff, a0=p.a0, b0=p.b0, Ncx=p.Ncx, Ncz=p.Ncz)


def _wrap(R, L):
    R[:, 0] -= L * np.rint(R[:, 0] / L)
    R[:, 1] -= L * np.rint(R[:, 1] / L)
    return R


def _state(O, case):
    """(R0 [3N], L, Lz, flags, kernel name expected, extra params)"""
    rs = np.random.RandomState(11)
    if case == "dense_film":          # fcc(16,4): ~46 pairs inside the cutoff per probe, 12 of 16 groups in reach, fewer
        L, Lz = 33.0, 240.0           # free lanes than wall sites -> fixed-lane fallback of assign_specials, several rounds
        R = O.fcc(16, 4, L=L).reshape(-1, 3).copy()
        R += 0.05 * rs.standard_normal(R.shape)
        return _wrap(R, L).ravel(), L, Lz, None, "mc64", {}
    if case == "dense_film_at_wall":  # the same pressed against the lower wall: wall sites act on most probes
        L, Lz = 33.0, 240.0
        R = O.fcc(16, 4, L=L).reshape(-1, 3).copy()
        R += 0.05 * rs.standard_normal(R.shape)
        R[:, 2] += -118.8 - R[:, 2].min()
        return _wrap(R, L).ravel(), L, Lz, None, "mc64", {}
    if case == "thin_film":           # fcc(24,1) in L = 48 (the widest box whose L/256 resolves the cutoff in 16 units):
        L, Lz = 48.0, 240.0           # thinner than the cutoff: every group in reach of every probe, both flag words
        R = O.fcc(24, 1, L=L).reshape(-1, 3).copy()
        R += 0.05 * rs.standard_normal(R.shape)
        return _wrap(R, L).ravel(), L, Lz, None, "mc64", {}
    if case == "two_slabs_ragged":    # N = 4000 (empty cells at the end of the z order), no walls, half of the film
        L, Lz = 33.0, 240.0           # 700 above the rest: most groups out of reach of every probe
        R = O.fcc(8, 16, L=L).reshape(-1, 3)[:4000].copy()
        R += 0.05 * rs.standard_normal(R.shape)
        R[R[:, 2] > 6.0, 2] += 700.0
        return _wrap(R, L).ravel(), L, Lz, "nowalls", "mc64", {}
    if case in ("unsafe_z_mb64", "unsafe_z_ma64"):
        # particles beyond the int16 range of the z words (zsafe = 32767 x 16 L/65536 = 264 at L = 33) are kept exact
        # through the `unsafe` masks, and a probe out there tests every real cell.  (sweep_kernel_mc* cannot get
        # there: its zsafe = 32766 L/256 lies beyond the upload bound |z| <= 4 Lz in every box it serves.)
        L, Lz = 33.0, 240.0
        R = O.fcc(8, 16, L=L).reshape(-1, 3)[:4000].copy()
        R += 0.05 * rs.standard_normal(R.shape)
        R[::131, 2] += 300.0          # 31 particles between 267 and 333
        R[7::400, 2] = 264.5 + 0.3 * np.arange(len(R[7::400]))          # ten just beyond zsafe, within the cutoff of each other
        return _wrap(R, L).ravel(), L, Lz, "nowalls", case[-4:], {"tune_kernel": S_KERNEL[case[-4:]]}
    if case in ("unsafe_z_mc32x4", "unsafe_z_mt64x8"):
        # Round 5: the several-wavefront byte-screen kernels have an unsafe z range INSIDE boxes they serve: zsafe = 32766 L/256 =
        # 128 L, the upload bound is |z| <= 4 Lz, and they are planned up to Lz < 250 L -- a narrow tall box (L = 12, Lz = 1200:
        # zsafe = 1536; L = 24, Lz = 2400: 3072) holds particles beyond it.  (The ONE-wavefront mc kernels cannot get there: their
        # plan needs sweep_kernel_ma's standard z unit below them, i.e. Lz < ~16 L, so 4 Lz < zsafe -- checked on the GPU: at
        # Lz = 55 L the plan gives sweep_kernel_mi, at 100 L sweep_kernel_mx.)  A film fcc(Na, 96 | 64), no walls, ~60-170 particles
        # 100-300 beyond zsafe and ten just beyond it within the cutoff of each other: unsafe particles are always candidates, an
        # unsafe probe tests every real cell of every wave, the cells' z words and the groups' ranges hold clamped values.
        Na, Nz, kern = {"unsafe_z_mc32x4": (4, 96, "mc32x4"), "unsafe_z_mt64x8": (8, 64, "mt64x8")}[case]
        L = 3.0 * Na
        Lz, zs = 100.0 * L, 128.0 * L
        R = O.fcc(Na, Nz, L=L, Lz=Lz).reshape(-1, 3).copy()
        R += 0.05 * rs.standard_normal(R.shape)
        far = R[::131]
        R[::131, 2] = zs + 100.0 + 3.0 * np.arange(len(far))
        R[7::400, 2] = zs + 1.0 + 0.3 * np.arange(len(R[7::400]))
        assert (np.abs(R[:, 2]) > zs).sum() > 40 and np.abs(R[:, 2]).max() < 4 * Lz
        return _wrap(R, L).ravel(), L, Lz, "nowalls", kern, {}
    if case == "resort_3":            # three sweeps per z sort: group ranges widened by two sweeps of accepted moves
        L, Lz = 33.0, 240.0
        return O.fcc(8, 16, L=L), L, Lz, None, "mc64", {"tune_resort": 3}
    if case in ("mc16_dense", "ml16_dense"):   # N = 1024 dense film through sweep_kernel_mc16 (asked for by name) and through
        L, Lz = 33.0, 240.0                    # sweep_kernel_ml16, which the plan gives few replicas of N <= 1024 (positions in LDS)
        R = O.fcc(16, 1, L=L).reshape(-1, 3).copy()
        R[:, 2] = 0.9 * rs.standard_normal(len(R))                      # a rough monolayer: overlaps, rejections
        R += 0.05 * rs.standard_normal(R.shape)
        return _wrap(R, L).ravel(), L, Lz, None, case[:4], ({"tune_kernel": 7} if case == "mc16_dense" else {})
    if case == "mc32_two_slabs":      # N = 2000 ragged, two slabs, through sweep_kernel_mc32
        L, Lz = 33.0, 240.0
        R = O.fcc(8, 8, L=L).reshape(-1, 3)[:2000].copy()
        R += 0.05 * rs.standard_normal(R.shape)
        R[R[:, 2] > 2.0, 2] += 60.0
        return _wrap(R, L).ravel(), L, Lz, None, "mc32", {}
    if case == "mc32x4_dense":        # 4096 < N <= 8192: the four-wavefront kernel on a dense film
        L, Lz = 33.0, 240.0
        R = O.fcc(16, 6, L=L).reshape(-1, 3).copy()                     # 6144 particles
        R += 0.05 * rs.standard_normal(R.shape)
        return _wrap(R, L).ravel(), L, Lz, None, "mc32x4", {}
    if case == "mc64x4_two_slabs":    # config 5's kernel with two separated slabs and a ragged N
        L, Lz = 33.0, 240.0
        R = O.fcc(12, 16, L=L).reshape(-1, 3)[:9000].copy()             # 9216 -> 9000
        R += 0.05 * rs.standard_normal(R.shape)
        R[R[:, 2] > 4.0, 2] += 55.0
        return _wrap(R, L).ravel(), L, Lz, None, "mc64x4", {}
    raise ValueError(case)


# the states of round 3's two-team kernel for N <= 1024 now run sweep_kernel_ml16 (which replaced it in the plan); the two-team
# kernel that remains, mt64x8, runs the states of the several-wavefront kernels it stands in for
def _tt_state(O, case):
    if case == "ml16_benchmark":
        return O.fcc(8, 4), 33.0, 240.0, None, "ml16", {}
    if case == "ml16_ragged_no_walls":
        rs = np.random.RandomState(5)
        R = O.fcc(8, 4).reshape(-1, 3)[:900].copy()
        R += 0.05 * rs.standard_normal(R.shape)
        R[R[:, 2] > 6.0, 2] += 40.0
        return _wrap(R, 33.0).ravel(), 33.0, 240.0, "nowalls", "ml16", {}
    if case == "ml16_at_wall":
        rs = np.random.RandomState(6)
        R = O.fcc(8, 4).reshape(-1, 3).copy()
        R += 0.05 * rs.standard_normal(R.shape)
        R[:, 2] += -118.6 - R[:, 2].min()
        return _wrap(R, 33.0).ravel(), 33.0, 240.0, None, "ml16", {}
    if case == "mt64x8_benchmark":
        return O.fcc(16, 16), 33.0, 240.0, None, "mt64x8", {}
    if case == "mt64x8_at_wall":       # the dense start pressed against the lower wall: wall sites, plane and side pair beside a
        rs = np.random.RandomState(8)  # hand-over list that fills the wavefront (second hand-over, lanes left with a third candidate)
        R = O.fcc(16, 16).reshape(-1, 3).copy()
        R += 0.03 * rs.standard_normal(R.shape)
        R[:, 2] += -118.9 - R[:, 2].min()
        return _wrap(R, 33.0).ravel(), 33.0, 240.0, None, "mt64x8", {}
    if case == "mt64x8_two_slabs":
        R0, L, Lz, mode, _, extra = _state(O, "mc64x4_two_slabs")
        return R0, L, Lz, mode, "mt64x8", extra
    if case == "mt64x8_unsafe_z":
        return _state(O, "unsafe_z_mt64x8")
    if case in ("mt64x8_condensed", "mt64x8_condensed_no_walls", "mt64x8_overfull", "mt64x8_overfull_no_walls"):
        # Round 5: a CONDENSED state, which is what a thermalised film becomes.  fcc(16,16) with the lattice
        # constant of the Lennard-Jones crystal (a = 1.6, rho = 0.98: L = 25.6; ~160 candidate bits per probe, 40 per wavefront)
        # and an OVERFULL one (a = 1.125, rho = 2.8, L = 18: ~100 candidate bits per wavefront and probe), where the first and
        # second hand-over of a wave fill its list: round 4's list length of 64 made s_bfm_b64's 6-bit count wrap to an EMPTY
        # mask on the waves without special lanes and dropped all 64 items (profiles/r05_two_team_list_overflow.txt: this case
        # against the round-4 generator).  With walls the film is pressed against the lower wall (wall lanes + side lanes in
        # front of the list on the slab-0 waves); without, every wave's list starts at lane 0.
        a = 1.6 if "condensed" in case else 1.125
        L, Lz = 16 * a, 240.0
        rs = np.random.RandomState(21)
        R = O.fcc(16, 16, L=L).reshape(-1, 3).copy()
        R += 0.03 * rs.standard_normal(R.shape)
        if case.endswith("no_walls"):
            R[:, 2] -= R[:, 2].mean()
        else:
            R[:, 2] += -118.9 - R[:, 2].min()
        # step size: with the reference's gamma = 1 (A = T, main.c:48-51) no move is ever accepted in a condensed state and the
        # comparison would be vacuous; A = 0.004 / 4e-5 gives acceptance ~0.4 (SMC.c:284, 307-313: A is a run-time argument)
        return (_wrap(R, L).ravel(), L, Lz, ("nowalls" if case.endswith("no_walls") else None), "mt64x8",
                {"A": 0.004 if "condensed" in case else 4e-5})
    raise ValueError(case)


TT_CASES = ["ml16_benchmark", "ml16_ragged_no_walls", "ml16_at_wall", "mt64x8_benchmark", "mt64x8_two_slabs", "mt64x8_at_wall",
            "mt64x8_condensed", "mt64x8_condensed_no_walls", "mt64x8_overfull", "mt64x8_overfull_no_walls", "mt64x8_unsafe_z"]

CASES = ["dense_film", "dense_film_at_wall", "thin_film", "two_slabs_ragged", "unsafe_z_mb64", "unsafe_z_ma64", "unsafe_z_mc32x4", "resort_3", "mc16_dense",
         "ml16_dense", "mc32_two_slabs", "mc32x4_dense", "mc64x4_two_slabs"]


@pytest.mark.parametrize("case", CASES + TT_CASES)
def test_rare_path_against_oracle(S, O, case):
    R0, L, Lz, mode, kernel, extra = _tt_state(O, case) if case in TT_CASES else _state(O, case)
    N = R0.size // 3
    nrep, eq, nsw = (8 if N <= 8192 else 4), 0, (3 if case == "resort_3" else 4 if N <= 1024 else 2 if N <= 2304 or case.startswith("mt64x8") else 1)
    flags = S.FLAG_SERIES | (S.FLAG_E0_RESTART if mode == "nowalls" else S.FLAGS_REFERENCE)
    geom = {"mc64": (64, 1), "mb64": (64, 1), "ma64": (64, 1), "mc32": (32, 1), "mc16": (16, 1), "ml16": (16, 1), "mc32x4": (0, 0),
            "mc64x4": (64, 4), "mt64x8": (64, 8)}[kernel]
    p = S.default_params(N, nrep, L=L, Lz=Lz, flags=flags, tune_slots=geom[0], tune_waves=geom[1], **extra)
    with S.Engine(p) as eng:
        assert eng.kernel_form[1] == "smcx::sweep_kernel_" + kernel, eng.kernel_form
        eng.upload(R0, O.W_FIXTURE)
        eng.run(eq, nsw, 1)
        E, jj = eng.series(nsw)
        ob = eng.observables()
        Rg = eng.positions()
    s = _sys(O, p)
    W = O.W_FIXTURE
    if mode == "nowalls":      # the oracle has no switch: a wall at infinity strength 0 = sites with zero strengths, plane off
        s = O.make_sys(N, M=p.M, L=L, Lz=Lz, cutoff=p.cutoff, a0=0.0, b0=0.0, Ncx=p.Ncx, Ncz=p.Ncz)
        W = np.zeros_like(O.W_FIXTURE)
    total = 0
    with ThreadPoolExecutor(min(nrep, len(os.sched_getaffinity(0)))) as ex:         # ctypes releases the GIL
        refs = list(ex.map(lambda r: O.chain(s, 12345 + r, R0, W, T, extra.get("A", A), eq, nsw, 1), range(nrep)))
    for r, ref in enumerate(refs):
        assert list(jj[r]) == list(ref["jj"]), (case, r, list(jj[r]), list(ref["jj"]))
        TOL.assert_series(E[r], ref["E"], what="%s replica %d" % (case, r))
        TOL.assert_positions(Rg[r], ref["R"], nsw, "%s replica %d" % (case, r))
        assert np.array_equal(ob["zhist"][r], ref["zhist"]), (case, r)
        total += int(ref["accepted"])
    assert total > 0 or case in ("dense_film", "dense_film_at_wall", "mc32x4_dense", "mt64x8_at_wall"), case


def test_benchmark_kernel_ensemble_statistics_against_the_oracle(S, O):
    """sweep_kernel_mc64 on the benchmark's system beyond the chaos horizon, with the ORACLE as the second arm:
    64 replicas x 40 sweeps each (the oracle: one chain per host core, ~35 s on 16 cores).  Chains separate after
    ~10 sweeps, so realised energies differ; the ensemble means of the final energy, of the accepted moves and the
    ensemble z profile must agree within 4 standard errors -- a bias from a rare path (group ranges, lane
    assignment, issue priorities are all configuration dependent) would accumulate here."""
    N, nrep, nsw = 4096, 64, 40
    R0 = O.fcc(8, 16)
    p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES)
    with S.Engine(p) as eng:
        assert eng.kernel_form[1] == "smcx::sweep_kernel_mc64"
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 10)
        E, jj = eng.series(nsw)
        zg = eng.observables()["zhist"].sum(axis=0).astype(float)
    s = _sys(O, p)

    def one(r):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, nsw, 10)
        return ref["E"][-1], float(ref["jj"].sum()), ref["zhist"].astype(float), ref["jj"][:5].copy(), ref["E"][:6].copy()
    with ThreadPoolExecutor(16) as ex:                      # ctypes releases the GIL
        refs = list(ex.map(one, range(nrep)))
    Eo = np.array([r[0] for r in refs]); jo = np.array([r[1] for r in refs])
    zo = np.sum([r[2] for r in refs], axis=0)
    # inside the horizon the chains are THE SAME chains (first five sweeps: equal accepted counts, E within the schedule)
    for r in range(nrep):
        assert list(jj[r][:5]) == list(refs[r][3]), r
        TOL.assert_series(E[r][:6], refs[r][4], what="replica %d" % r)
    for what, a, b in (("final energy", E[:, -1], Eo), ("accepted moves", jj.sum(axis=1).astype(float), jo)):
        se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        print("%s: GPU %.4f vs oracle %.4f (difference %.2f standard errors)" % (what, a.mean(), b.mean(), (a.mean() - b.mean()) / se))
        assert abs(a.mean() - b.mean()) < 4 * se, what
    # the wall-normal profile of the ensemble (4 gathers x 64 replicas x 4096 particles): bins within 4 sigma (Poisson)
    assert zg.sum() == zo.sum()
    dev = np.abs(zg - zo) / np.sqrt(np.maximum(zg + zo, 1.0))
    assert dev.max() < 4.5, dev.max()
    assert (jj.sum(axis=1) != jo).any()                     # and they did separate: the comparison is not vacuous
