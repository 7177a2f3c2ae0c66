"""montecarlo-surfacer_amd -- MI355X-native Smart-Monte-Carlo engine (Python glue).

The product is the C-ABI shared library ``libsmcx.so`` (HIP kernels for gfx950 +
``include/smcx.h``); this module is only a ctypes binding of that ABI used by the
tests, ``bench.py`` and the multi-GPU driver.  There is no CPU fallback: if the
library is missing this import raises, and without a GPU every compute call
returns an error status which is raised as :class:`SmcxError`.

The directory name has a hyphen (it mirrors the reference's name), so import it
through ``smcx_loader.load()`` at the repo root or with importlib.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMCX_LIB") or os.path.join(_HERE, "libsmcx.so")  # SMCX_LIB: diagnostic builds

OK, ERR_PARAM, ERR_HIP, ERR_STATE, ERR_NOMEM, ERR_UNSUPPORTED, ERR_NODEVICE, ERR_RCCL = range(8)
FLAG_WALLS, FLAG_E0_RESTART, FLAG_SERIES, FLAG_FULL_HIST, FLAG_PRESSURE, FLAG_CLUSTERS = 1, 2, 4, 8, 16, 32
FLAGS_REFERENCE = FLAG_WALLS | FLAG_E0_RESTART
OBS_RECORD_DOUBLES = 8
KERNEL_AUTO, KERNEL_FP64, KERNEL_SCREENED, KERNEL_MX, KERNEL_MI, KERNEL_MA, KERNEL_MB, KERNEL_MC, KERNEL_MT = range(9)


class Params(C.Structure):
    """mirror of smcx_params (include/smcx.h)"""
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("nrep", C.c_int32), ("device", C.c_int32),
                ("L", C.c_double), ("Lz", C.c_double), ("T", C.c_double), ("A", C.c_double),
                ("cutoff", C.c_double), ("a0", C.c_double), ("b0", C.c_double),
                ("Ncx", C.c_int32), ("Ncz", C.c_int32), ("flags", C.c_uint32),
                ("base_seed", C.c_uint32), ("first_replica", C.c_uint32),
                ("tune_slots", C.c_int32), ("tune_waves", C.c_int32),
                ("lca_time", C.c_int32), ("tune_kernel", C.c_int32), ("tune_resort", C.c_int32),
                ("lca_cutoff", C.c_double)]


class SmcxError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: status %d (%s)" % (where, status, _lib().smcx_strerror(status).decode())
        if detail:
            msg += " -- " + detail
        super().__init__(msg)


_LIB = None
_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)

EXPORTS = [
    "smcx_default_params", "smcx_device_count", "smcx_create", "smcx_destroy", "smcx_strerror",
    "smcx_last_error_string", "smcx_upload", "smcx_run", "smcx_observables",
    "smcx_therm_acceptance", "smcx_hist_info", "smcx_series", "smcx_density",
    "smcx_pressure_series", "smcx_acf", "smcx_download_positions",
    "smcx_total_energy", "smcx_rng_export", "smcx_rng_import", "smcx_obs_device_bytes",
    "smcx_export_observables_device", "smcx_last_kernel_ms", "smcx_last_run_ms", "smcx_geometry", "smcx_eval_moves",
    "smcx_rng_seed", "smcx_one_particle_moves",
    "smcx_cluster_counts", "smcx_cluster_update", "smcx_cluster_analysis", "smcx_kernel_form", "smcx_screen_bound",
    "smcx_screen_bound_int", "smcx_screen_bound_byte", "smcx_last_clock", "smcx_debug_wave_spread",
    "smcx_debug_clk_rows", "smcx_kernel_source_id", "smcx_replica_granule", "smcx_debug_window_schedule",
]
HOST_EXPORTS = ["smcx_host_sMC", "smcx_host_sMC_multi", "smcx_host_multi_error", "smcx_host_sim_free", "smcx_host_fcc_init",
                "smcx_host_initialize_box", "smcx_host_initialize_walls", "smcx_host_box_for_N", "smcx_host_write_csv",
                "smcx_host_read_last_state", "smcx_host_srand", "smcx_host_rand", "smcx_host_vec_box_muller",
                "smcx_host_nowall_energy_single", "smcx_host_nowall_force", "smcx_host_nowall_energy",
                "smcx_host_nowall_pressure", "smcx_host_nowall_fcc", "smcx_host_nowall_sweep", "smcx_host_nowall_sMC",
                "smcx_host_nowall_box"]


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsmcx.so is not built (run __graft_entry__.build() or "
                              "make -C montecarlo-surfacer_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.smcx_default_params.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32]
        L.smcx_default_params.restype = None
        L.smcx_device_count.argtypes = [C.POINTER(C.c_int)]
        L.smcx_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
        L.smcx_destroy.argtypes = [vp]
        L.smcx_strerror.argtypes = [C.c_int]
        L.smcx_strerror.restype = C.c_char_p
        L.smcx_last_error_string.argtypes = [vp]
        L.smcx_last_error_string.restype = C.c_char_p
        L.smcx_upload.argtypes = [vp, _dp, C.c_int, _dp, _u32p]
        L.smcx_run.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.smcx_observables.argtypes = [vp, _dp, _dp, _dp, _u64p, _u64p, _dp]
        L.smcx_therm_acceptance.argtypes = [vp, _dp]
        L.smcx_hist_info.argtypes = [vp, _u64p, _u64p]
        L.smcx_series.argtypes = [vp, _dp, _i32p]
        L.smcx_density.argtypes = [vp, _u64p, _u64p]
        L.smcx_pressure_series.argtypes = [vp, _dp, C.POINTER(C.c_int)]
        L.smcx_cluster_counts.argtypes = [vp, _u64p, _u64p, _u64p, _u64p, C.POINTER(C.c_int)]
        L.smcx_cluster_update.argtypes = [vp]
        L.smcx_cluster_analysis.argtypes = [vp, C.c_int, _i32p, _u64p]
        L.smcx_acf.argtypes = [vp, C.c_int, _dp, C.POINTER(C.c_int), _dp, _dp]
        L.smcx_download_positions.argtypes = [vp, _dp]
        L.smcx_total_energy.argtypes = [vp, _dp]
        L.smcx_rng_export.argtypes = [vp, _u32p]
        L.smcx_rng_import.argtypes = [vp, _u32p]
        L.smcx_obs_device_bytes.argtypes = [vp]
        L.smcx_obs_device_bytes.restype = C.c_size_t
        L.smcx_export_observables_device.argtypes = [vp, vp, C.c_size_t]
        L.smcx_last_kernel_ms.argtypes = [vp, _dp, C.POINTER(C.c_int)]
        L.smcx_last_run_ms.argtypes = [vp, _dp]
        L.smcx_last_clock.argtypes = [vp, _dp, _dp]
        L.smcx_debug_wave_spread.argtypes = [vp, _dp]
        L.smcx_debug_clk_rows.argtypes = [vp, _u64p]
        L.smcx_geometry.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.smcx_kernel_form.argtypes = [vp, C.POINTER(C.c_int), C.c_char_p, C.c_int]
        L.smcx_screen_bound.argtypes = [C.POINTER(Params), C.c_int, _dp, _dp, _dp, _dp]
        L.smcx_screen_bound_int.argtypes = [C.POINTER(Params), _dp, _dp, _dp, _dp, _dp, _i32p, _i32p]
        L.smcx_eval_moves.argtypes = [C.POINTER(Params), _dp, _dp, _i32p, _dp, _dp]
        L.smcx_rng_seed.argtypes = [_u32p, C.c_uint32]
        L.smcx_rng_seed.restype = None
        L.smcx_one_particle_moves.argtypes = [C.POINTER(Params), _u32p, _dp, _dp, _dp, C.c_double,
                                              C.c_double, C.POINTER(C.c_int), _dp]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def default_params(N, nrep, **kw):
    p = Params()
    _lib().smcx_default_params(C.byref(p), N, nrep)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def kernel_source_id(kernel):
    """source identity (16 hex digits) of a sweep kernel of the loaded library (smcx_kernel_source_id); None if unknown"""
    buf = C.create_string_buffer(32)
    f = _lib().smcx_kernel_source_id
    f.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    f.restype = C.c_int
    return buf.value.decode() if f(kernel.encode(), buf, 32) == OK else None


def window_schedule(nrep, granule, nsweeps, every, max_launches=4096):
    """the launches of one launch group (smcx_debug_window_schedule): rows of (workgroups, u0, nmod, blk0, window, sw0, nsw, sw0', nsw')"""
    out = np.zeros((max_launches, 9), dtype=np.int32)
    f = _lib().smcx_debug_window_schedule
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]
    f.restype = C.c_int
    n = f(nrep, granule, nsweeps, every, _p(out, C.c_int32), max_launches)
    if n < 0:
        raise SmcxError(-n, "smcx_debug_window_schedule")
    return out[:min(n, max_launches)], n


def device_count():
    n = C.c_int(0)
    _lib().smcx_device_count(C.byref(n))
    return n.value


def geometry_supported_fp64(slots, waves):
    """geometries the fp64 sweep kernels are built for (csrc/smcx_kernels.hip: lookup)"""
    return (slots, waves) in {(1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1), (64, 1), (8, 2), (16, 2),
                              (32, 2), (4, 4), (8, 4), (16, 4), (32, 4), (8, 8), (16, 8), (32, 8),
                              (16, 16), (32, 16)}


def screen_bound(p, lds_z):
    """(thr, u2, to_fixed, zsafe) of the screened kernel's cutoff test for the box of p"""
    v = [C.c_double() for _ in range(4)]
    rc = _lib().smcx_screen_bound(C.byref(p), int(lds_z), *[C.byref(x) for x in v])
    if rc != OK:
        raise SmcxError(rc, "smcx_screen_bound")
    return tuple(x.value for x in v)


def screen_bound_int(p):
    """(thr, u2, to_fixed, zsafe, uz, neg_c, zshift) of the integer screen (sweep_kernel_mi) for the box of p"""
    v = [C.c_double() for _ in range(5)]
    nc, zs = C.c_int32(), C.c_int32()
    rc = _lib().smcx_screen_bound_int(C.byref(p), *[C.byref(x) for x in v], C.byref(nc), C.byref(zs))
    if rc != OK:
        raise SmcxError(rc, "smcx_screen_bound_int")
    return tuple(x.value for x in v) + (nc.value, zs.value)


def screen_bound_byte(p):
    """(to_fixed, zsafe, neg_t, reach_z) of the byte screen (sweep_kernel_mc64) for the box of p"""
    tf, zs = C.c_double(), C.c_double()
    nt, rz = C.c_int32(), C.c_int32()
    rc = _lib().smcx_screen_bound_byte(C.byref(p), C.byref(tf), C.byref(zs), C.byref(nt), C.byref(rz))
    if rc != OK:
        raise SmcxError(rc, "smcx_screen_bound_byte")
    return tf.value, zs.value, nt.value, rz.value


def rng_seed(seed):
    st = np.zeros(32, dtype=np.uint32)
    _lib().smcx_rng_seed(_p(st, C.c_uint32), seed)
    return st


def eval_moves(params, R, W, n, prop):
    """teacher-forced Um,Fm,Un,Fn (SMC.c:300-304, 319-321); returns [nrep][8]"""
    R = np.ascontiguousarray(R, dtype=np.float64)
    n = np.ascontiguousarray(n, dtype=np.int32)
    prop = np.ascontiguousarray(prop, dtype=np.float64)
    W = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
    out = np.zeros((params.nrep, 8))
    rc = _lib().smcx_eval_moves(C.byref(params), _p(R, C.c_double), _p(W, C.c_double),
                                _p(n, C.c_int32), _p(prop, C.c_double), _p(out, C.c_double))
    if rc != OK:
        raise SmcxError(rc, "smcx_eval_moves", _lib().smcx_last_error_string(None).decode())
    return out


def one_particle_moves(params, rng, R, Rn, W, A, T, j, U):
    """oneParticleMoves (SMC.h:102) for one chain on the GPU; returns (j, U), arrays in place"""
    jj = C.c_int(j)
    UU = C.c_double(U)
    W = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
    rc = _lib().smcx_one_particle_moves(C.byref(params), _p(rng, C.c_uint32), _p(R, C.c_double),
                                        _p(Rn, C.c_double), _p(W, C.c_double), A, T, C.byref(jj),
                                        C.byref(UU))
    if rc != OK:
        raise SmcxError(rc, "smcx_one_particle_moves", _lib().smcx_last_error_string(None).decode())
    return jj.value, UU.value


class Engine:
    """One handle = the replica chains of one GPU (the batched form of sMC, SMC.c:21)."""

    def __init__(self, params):
        self.p = params
        self._h = C.c_void_p()
        self._chk(_lib().smcx_create(C.byref(params), C.byref(self._h)), "smcx_create")

    def _chk(self, rc, where):
        if rc != OK:
            detail = _lib().smcx_last_error_string(self._h).decode() if self._h else \
                _lib().smcx_last_error_string(None).decode()
            raise SmcxError(rc, where, detail)

    def close(self):
        if self._h:
            _lib().smcx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def geometry(self):
        s, w, l = C.c_int(), C.c_int(), C.c_int()
        self._chk(_lib().smcx_geometry(self._h, C.byref(s), C.byref(w), C.byref(l)), "smcx_geometry")
        return s.value, w.value, l.value

    @property
    def kernel_form(self):
        """(form, name): 1 = fp64 sweep kernel, 2 = screened sweep kernel"""
        f = C.c_int(0)
        buf = C.create_string_buffer(96)
        self._chk(_lib().smcx_kernel_form(self._h, C.byref(f), buf, 96), "smcx_kernel_form")
        return f.value, buf.value.decode()

    def upload(self, R0, W, seeds=None):
        R0 = np.ascontiguousarray(R0, dtype=np.float64)
        per = 1 if R0.size == self.p.nrep * 3 * self.p.N and self.p.nrep > 1 else 0
        if not per and R0.size != 3 * self.p.N:
            raise ValueError("R0 must be [3N] or [nrep][3N]")
        W = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
        seeds = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        self._chk(_lib().smcx_upload(self._h, _p(R0, C.c_double), per, _p(W, C.c_double),
                                     _p(seeds, C.c_uint32)), "smcx_upload")

    def run(self, eqsteps, maxsteps, gather_lapse):
        self._chk(_lib().smcx_run(self._h, eqsteps, maxsteps, gather_lapse), "smcx_run")

    def observables(self):
        n, ncz = self.p.nrep, self.p.Ncz
        acc, mE, dE, El = (np.zeros(n) for _ in range(4))
        zh = np.zeros((n, ncz), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint64)
        self._chk(_lib().smcx_observables(self._h, _p(acc, C.c_double), _p(mE, C.c_double),
                                          _p(dE, C.c_double), _p(zh, C.c_uint64),
                                          _p(cnt, C.c_uint64), _p(El, C.c_double)), "smcx_observables")
        return dict(acceptance_ratio=acc, meanE=mE, dE=dE, zhist=zh, accepted=cnt, E_last=El)

    def therm_acceptance(self):
        r = np.zeros(self.p.nrep)
        self._chk(_lib().smcx_therm_acceptance(self._h, _p(r, C.c_double)), "smcx_therm_acceptance")
        return r

    def hist_info(self):
        g = np.zeros(self.p.nrep, dtype=np.uint64)
        o = np.zeros(self.p.nrep, dtype=np.uint64)
        self._chk(_lib().smcx_hist_info(self._h, _p(g, C.c_uint64), _p(o, C.c_uint64)), "smcx_hist_info")
        return g, o

    def series(self, maxsteps):
        E = np.zeros((self.p.nrep, maxsteps + 1))
        jj = np.zeros((self.p.nrep, max(maxsteps, 1)), dtype=np.int32)
        self._chk(_lib().smcx_series(self._h, _p(E, C.c_double), _p(jj, C.c_int32)), "smcx_series")
        return E, jj[:, :maxsteps]

    def density(self):
        Nc = self.p.Ncx * self.p.Ncx * self.p.Ncz
        D = np.zeros((self.p.nrep, Nc), dtype=np.uint64)
        Mu = np.zeros((self.p.nrep, Nc), dtype=np.uint64)
        self._chk(_lib().smcx_density(self._h, _p(D, C.c_uint64), _p(Mu, C.c_uint64)), "smcx_density")
        return D, Mu

    def pressure_series(self):
        n = C.c_int(0)
        self._chk(_lib().smcx_pressure_series(self._h, None, C.byref(n)), "smcx_pressure_series")
        P = np.zeros((self.p.nrep, max(n.value, 1)))
        self._chk(_lib().smcx_pressure_series(self._h, _p(P, C.c_double), C.byref(n)), "smcx_pressure_series")
        return P[:, :n.value]

    def cluster_counts(self):
        """counts behind l1, l2[], l3[] of SMC.c:146-155: (n1[nrep], h2[nrep][16], h3[nrep][16],
        overflow[nrep], analyses)"""
        n = self.p.nrep
        n1 = np.zeros(n, dtype=np.uint64); ov = np.zeros(n, dtype=np.uint64)
        h2 = np.zeros((n, 16), dtype=np.uint64); h3 = np.zeros((n, 16), dtype=np.uint64)
        k = C.c_int(0)
        self._chk(_lib().smcx_cluster_counts(self._h, _p(n1, C.c_uint64), _p(h2, C.c_uint64),
                                             _p(h3, C.c_uint64), _p(ov, C.c_uint64), C.byref(k)),
                  "smcx_cluster_counts")
        return n1, h2, h3, ov, k.value

    def cluster_update(self):
        self._chk(_lib().smcx_cluster_update(self._h), "smcx_cluster_update")

    def cluster_analysis(self, replica=0):
        """clusterAnalysis (SMC.c:971-1045) of one replica: (LCA[N(N-1)/2][3], overflow)"""
        npairs = self.p.N * (self.p.N - 1) // 2
        LCA = np.zeros((npairs, 3), dtype=np.int32)
        ov = np.zeros(1, dtype=np.uint64)
        self._chk(_lib().smcx_cluster_analysis(self._h, replica, _p(LCA, C.c_int32), _p(ov, C.c_uint64)),
                  "smcx_cluster_analysis")
        return LCA, int(ov[0])

    def acf(self, k_max=2500000):
        """fft_acf of the energy series (SMC.c:1051-1089): (acf[nrep][k], tau[nrep], cv[nrep])"""
        k = C.c_int(0)
        maxsteps_plus1 = None
        # k_eff depends only on the series length; ask with a throw-away buffer sized for the worst case
        tau = np.zeros(self.p.nrep); cv = np.zeros(self.p.nrep)
        rc = _lib().smcx_acf(self._h, k_max, None, C.byref(k), _p(tau, C.c_double), _p(cv, C.c_double))
        self._chk(rc, "smcx_acf")
        acf = np.zeros((self.p.nrep, k.value))
        self._chk(_lib().smcx_acf(self._h, k_max, _p(acf, C.c_double), C.byref(k), _p(tau, C.c_double),
                                  _p(cv, C.c_double)), "smcx_acf")
        return acf, tau, cv

    def positions(self):
        R = np.zeros((self.p.nrep, 3 * self.p.N))
        self._chk(_lib().smcx_download_positions(self._h, _p(R, C.c_double)), "smcx_download_positions")
        return R

    def total_energy(self):
        E = np.zeros(self.p.nrep)
        self._chk(_lib().smcx_total_energy(self._h, _p(E, C.c_double)), "smcx_total_energy")
        return E

    def rng_export(self):
        st = np.zeros((self.p.nrep, 32), dtype=np.uint32)
        self._chk(_lib().smcx_rng_export(self._h, _p(st, C.c_uint32)), "smcx_rng_export")
        return st

    def rng_import(self, st):
        st = np.ascontiguousarray(st, dtype=np.uint32)
        self._chk(_lib().smcx_rng_import(self._h, _p(st, C.c_uint32)), "smcx_rng_import")

    def obs_device_bytes(self):
        return _lib().smcx_obs_device_bytes(self._h)

    def export_observables_device(self, dev_ptr, nbytes):
        self._chk(_lib().smcx_export_observables_device(self._h, C.c_void_p(dev_ptr), nbytes),
                  "smcx_export_observables_device")

    def replica_granule(self):
        """(replicas the device runs at once with this kernel, advisory text or "")"""
        g = C.c_int(0)
        buf = C.create_string_buffer(640)
        f = _lib().smcx_replica_granule
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_int]
        self._chk(f(self._h, C.byref(g), buf, 640), "smcx_replica_granule")
        return g.value, buf.value.decode()

    def last_run_ms(self):
        ms = C.c_double()
        self._chk(_lib().smcx_last_run_ms(self._h, C.byref(ms)), "smcx_last_run_ms")
        return ms.value

    def last_clock(self):
        """(GHz, wavefront lifetime in cycles) of the last sweep launch, measured in the kernel"""
        g, c = C.c_double(), C.c_double()
        self._chk(_lib().smcx_last_clock(self._h, C.byref(g), C.byref(c)), "smcx_last_clock")
        return g.value, c.value

    def wave_spread(self):
        """(min, median, max, span) of the wavefront lifetimes of the last sweep launch, microseconds"""
        out = (C.c_double * 4)()
        self._chk(_lib().smcx_debug_wave_spread(self._h, out), "smcx_debug_wave_spread")
        return tuple(out)

    def clk_rows(self, cols=4):
        """cols = 32 with the two-team stamps variant of the library (tools/probes/tt_phases.py)"""
        out = np.zeros((self.p.nrep, cols), dtype=np.uint64)
        self._chk(_lib().smcx_debug_clk_rows(self._h, _p(out, C.c_uint64)), "smcx_debug_clk_rows")
        return out

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_int()
        self._chk(_lib().smcx_last_kernel_ms(self._h, C.byref(ms), C.byref(n)), "smcx_last_kernel_ms")
        return ms.value, n.value


# ---- C host side (libsmcx_host.so): system preparation, include/smcx_host.h -------
HOST_LIB_PATH = os.path.join(_HERE, "libsmcx_host.so")
_HOST = None

# wall strengths the reference's initializeWalls(1.6, 0.0, 3.0, 0.5) produces on glibc
# (SMC.c:475-501, main.c:74-87); equal to host_initialize_walls(M=3, uninit=0.0)
W_REFERENCE = np.array([
    962.2264072645321, 57.35316319850277, 874.39446992695275, 52.11797177356199,
    857.36680597299653, 51.103043912231705, 1024.1964124687327, 61.046863345428257,
    925.40789594507817, 55.158608910148025, 913.63518965684239, 54.456900933792724,
    848.90539177252572, 50.598704324515197, 992.35137245273086, 59.148751047416368,
    844.42493013196849, 50.331648000000015])


class HostSim(C.Structure):
    """mirror of smcx_sim (include/smcx_host.h)"""
    _fields_ = [("nrep", C.c_int), ("N", C.c_int), ("Ncz", C.c_int),
                ("E", C.c_double), ("dE", C.c_double), ("acceptance_ratio", C.c_double),
                ("therm_acceptance", C.c_double),
                ("rep_E", _dp), ("rep_dE", _dp), ("rep_acceptance", _dp), ("zprofile", _dp), ("Rfinal", _dp),
                ("l1", C.c_double), ("l2", C.c_double * 16), ("l3", C.c_double * 16), ("lca_analyses", C.c_int),
                ("P", C.c_double), ("dP", C.c_double), ("tau", C.c_double), ("cv", C.c_double),
                ("kernel_ms", C.c_double), ("pair_evals_per_s", C.c_double)]


def _host():
    global _HOST
    if _HOST is None:
        _lib()
        H = C.CDLL(HOST_LIB_PATH)
        H.smcx_host_fcc_init.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        H.smcx_host_initialize_box.argtypes = [C.c_double, C.c_double, C.c_int, _dp]
        H.smcx_host_initialize_walls.argtypes = [C.c_double] * 4 + [C.c_int, C.c_double, _dp]
        H.smcx_host_initialize_walls.restype = None
        H.smcx_host_box_for_N.argtypes = [C.c_int, _dp, _dp]
        H.smcx_host_box_for_N.restype = None
        H.smcx_host_write_csv.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_char_p]
        H.smcx_host_read_last_state.argtypes = [C.c_char_p, C.c_int, _dp]
        H.smcx_host_sMC.argtypes = [C.POINTER(Params), _dp, _dp, C.c_int, C.c_int, C.c_int, C.POINTER(HostSim)]
        H.smcx_host_sMC_multi.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_int), _dp, _dp, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(HostSim)]
        H.smcx_host_multi_error.restype = C.c_char_p
        H.smcx_host_nowall_energy_single.argtypes = [C.c_int, _dp, C.c_double, C.c_int]
        H.smcx_host_nowall_energy_single.restype = C.c_double
        H.smcx_host_nowall_force.argtypes = [C.c_int, _dp, C.c_double, C.c_int, _dp]
        H.smcx_host_nowall_force.restype = None
        H.smcx_host_nowall_energy.argtypes = [C.c_int, _dp, C.c_double]
        H.smcx_host_nowall_energy.restype = C.c_double
        H.smcx_host_nowall_pressure.argtypes = [C.c_int, _dp, C.c_double]
        H.smcx_host_nowall_pressure.restype = C.c_double
        H.smcx_host_nowall_fcc.argtypes = [C.c_int, C.c_double, _dp]
        H.smcx_host_nowall_sMC.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint, C.c_int, C.c_int, _dp, _dp, _dp,
                                           _i32p]
        H.smcx_host_sim_free.argtypes = [C.POINTER(HostSim)]
        H.smcx_host_sim_free.restype = None
        _HOST = H
    return _HOST


def fcc_init(Na, Nz, L=33.0, Lz=240.0):
    """fcc(Na,Nz) slab start of SURVEY.md 8d (lattice code of SMC.c:432-461)"""
    X = np.zeros(12 * Na * Na * Nz)
    n = _host().smcx_host_fcc_init(Na, Nz, L, Lz, _p(X, C.c_double))
    if n != 4 * Na * Na * Nz:
        raise ValueError("bad lattice")
    return X


def initialize_box(N, L, Lz):
    """the reference's initializeBox (SMC.c:413-465); returns (X, particles placed)"""
    X = np.zeros(3 * N)
    return X, _host().smcx_host_initialize_box(L, Lz, N, _p(X, C.c_double))


def initialize_walls(M=3, x0m=1.6, x0sigma=0.0, ymm=3.0, ymsigma=0.5, uninit=0.0):
    W = np.zeros(2 * M * M)
    _host().smcx_host_initialize_walls(x0m, x0sigma, ymm, ymsigma, M, uninit, _p(W, C.c_double))
    return W


def write_csv(engine, maxsteps, gather_lapse, directory):
    """the reference's data_/local_/last_state_ files for the engine's last run (C host side)"""
    rc = _host().smcx_host_write_csv(engine._h, C.byref(engine.p), maxsteps, gather_lapse,
                                     directory.encode())
    if rc != OK:
        raise SmcxError(rc, "smcx_host_write_csv")


def host_sMC(p, W, R0, maxsteps, gather_lapse, eqsteps, gpus=0, devices=None):
    """smcx_host_sMC (the C driver mirroring sMC, SMC.c:21-267) -> dict of its results; gpus > 0:
    smcx_host_sMC_multi, the replicas dealt over that many devices and the observables gathered by RCCL"""
    sim = HostSim()
    W = np.ascontiguousarray(W, dtype=np.float64)
    R0 = np.ascontiguousarray(R0, dtype=np.float64)
    if gpus:
        dv = None if devices is None else (C.c_int * gpus)(*devices)
        rc = _host().smcx_host_sMC_multi(C.byref(p), gpus, dv, _p(W, C.c_double), _p(R0, C.c_double), maxsteps,
                                         gather_lapse, eqsteps, C.byref(sim))
        if rc != OK:
            raise SmcxError(rc, "smcx_host_sMC_multi", _host().smcx_host_multi_error().decode())
    else:
        rc = _host().smcx_host_sMC(C.byref(p), _p(W, C.c_double), _p(R0, C.c_double), maxsteps, gather_lapse,
                                   eqsteps, C.byref(sim))
        if rc != OK:
            raise SmcxError(rc, "smcx_host_sMC")
    n = sim.nrep
    out = {k: getattr(sim, k) for k in ("E", "dE", "acceptance_ratio", "therm_acceptance", "l1", "lca_analyses",
                                        "P", "dP", "tau", "cv", "kernel_ms", "pair_evals_per_s")}
    out["l2"] = np.array(sim.l2[:]); out["l3"] = np.array(sim.l3[:])
    out["rep_E"] = np.ctypeslib.as_array(sim.rep_E, (n,)).copy()
    out["rep_acceptance"] = np.ctypeslib.as_array(sim.rep_acceptance, (n,)).copy()
    out["zprofile"] = np.ctypeslib.as_array(sim.zprofile, (sim.Ncz,)).copy()
    out["Rfinal"] = np.ctypeslib.as_array(sim.Rfinal, (n, 3 * sim.N)).copy()
    _host().smcx_host_sim_free(C.byref(sim))
    return out


class NoWall:
    """BASELINE config 1 (SMC_noMPI_noWall.c): one chain on the host CPU, C host library (host/smcx_host_nowall.c)"""

    def __init__(self, N, rho):
        self.N, self.rho, self.L = N, rho, float(np.cbrt(N / rho))

    def fcc(self):
        X = np.zeros(3 * self.N)
        placed = _host().smcx_host_nowall_fcc(self.N, self.L, _p(X, C.c_double))
        return X, placed

    def energy(self, R):
        return _host().smcx_host_nowall_energy(self.N, _p(R, C.c_double), self.L)

    def pressure(self, R):
        return _host().smcx_host_nowall_pressure(self.N, _p(R, C.c_double), self.L)

    def single(self, R, i):
        F = np.zeros(3)
        e = _host().smcx_host_nowall_energy_single(self.N, _p(R, C.c_double), self.L, int(i))
        _host().smcx_host_nowall_force(self.N, _p(R, C.c_double), self.L, int(i), _p(F, C.c_double))
        return e, F

    def sMC(self, R, T, A, seed, maxsteps, gather_lapse=1):
        """runs the chain in place on R; returns (E[k], P[k], jj[n])"""
        ng = (maxsteps + gather_lapse - 1) // gather_lapse
        E = np.zeros(max(ng, 1)); P = np.zeros(max(ng, 1)); jj = np.zeros(max(maxsteps, 1), dtype=np.int32)
        rc = _host().smcx_host_nowall_sMC(self.N, self.L, T, A, int(seed), maxsteps, gather_lapse, _p(R, C.c_double),
                                          _p(E, C.c_double), _p(P, C.c_double), _p(jj, C.c_int32))
        if rc != OK:
            raise SmcxError(rc, "smcx_host_nowall_sMC")
        return E[:ng], P[:ng], jj[:maxsteps]


def read_last_state(path, N):
    R = np.zeros(3 * N)
    n = _host().smcx_host_read_last_state(path.encode(), N, _p(R, C.c_double))
    if n != 3 * N:
        raise ValueError("last_state file holds %d of %d coordinates" % (n, 3 * N))
    return R
