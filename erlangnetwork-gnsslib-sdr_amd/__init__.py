"""Host-side Python mirror of libgnsscorr.so (MI355X GNSS correlation engine).

The product is the C-ABI shared library built from ``csrc/`` (HIP kernels for
gfx950 + C host code); this package only binds it with ctypes so that tests,
``bench.py`` and ``__graft_entry__.py`` can drive it, and mirrors the
reference's channel set-up (``initsdrch``, ref src/sdrinit.c:583-657) so that
call sites read like the reference's.  There is no CPU fallback: importing
works without a GPU, but every compute entry point fails loudly when the
library or a device is missing.

The directory name contains '-', so it is loaded through ``gnsscorr_loader``
(repo root) under the module name ``erlangnetwork_gnsslib_sdr_amd``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNSSCORR_LIB", os.path.join(_HERE, "libgnsscorr.so"))   # override: A/B builds

MAXTAPS = 33
FMT_STEREO, FMT_RTLSDR = 1, 2
CTYPE_L1CA, CTYPE_G1, CTYPE_L1SBAS = 1, 20, 27
SYS_GPS, SYS_SBS, SYS_GLO = 1, 2, 4
DTYPEI, DTYPEIQ = 1, 2
FTYPE1, FTYPE2 = 1, 2
MEMBUFFLEN, FILE_BUFFSIZE = 5000, 65536      # ref src/sdr.h:134,137
ACQTH = 3.0                                  # ref src/sdr.h:148


class GnsscorrError(RuntimeError):
    pass


# ---- ctypes mirrors of include/gnsscorr.h ---------------------------------
class ChanDesc(C.Structure):
    _fields_ = [("prn", C.c_int), ("ctype", C.c_int), ("dtype", C.c_int), ("ftype", C.c_int),
                ("clen", C.c_int), ("nsamp", C.c_int), ("nsampchip", C.c_int),
                ("f_sf", C.c_double), ("f_if", C.c_double), ("foffset", C.c_double),
                ("crate", C.c_double), ("ctime", C.c_double), ("ti", C.c_double),
                ("code", C.POINTER(C.c_short)), ("intg", C.c_int), ("nfreq", C.c_int),
                ("freq", C.POINTER(C.c_double)), ("nfft", C.c_int), ("corrn", C.c_int),
                ("corrp", C.POINTER(C.c_int))]


class TrkState(C.Structure):
    _fields_ = [("carrfreq", C.c_double), ("codefreq", C.c_double), ("remcode", C.c_double),
                ("remcarr", C.c_double), ("buffloc", C.c_uint64)]




class LoopState(C.Structure):
    """gnsscorr_loop_t (include/gnsscorr.h): loop-filter configuration and state of one channel."""
    _fields_ = ([(n, C.c_double) for n in ("acqfreq", "f_if", "foffset", "f_cf", "crate", "ctime")] +
                [(n, C.c_double * 2) for n in ("pllaw", "pllw2", "fllw", "dllaw", "dllw2")] +
                [(n, C.c_int) for n in ("ne", "nl", "loopms", "rate", "flagsync", "synci", "navcnt", "swloop")] +
                [("cnt", C.c_uint64)] +
                [(n, C.c_double) for n in ("carrNco", "codeNco", "carrErr", "codeErr", "freqErr")] +
                [(n, C.c_double * MAXTAPS) for n in ("II", "QQ", "oldI", "oldQ", "sumI", "sumQ", "oldsumI", "oldsumQ")] +
                [(n, C.c_int) for n in ("prn", "biti", "bit", "swsync", "swreset", "flagpol")] +
                [("bitIP", C.c_double), ("bitsync", C.c_int * 20)])


class TrkLog(C.Structure):
    """gnsscorr_trklog_t: one row per code period of a closed-loop run."""
    _fields_ = ([(n, C.c_double) for n in ("carrfreq", "codefreq", "carrErr", "codeErr", "carrNco", "codeNco", "freqErr",
                                            "remcode", "remcarr")] +
                [("buffloc", C.c_uint64), ("currnsamp", C.c_int), ("flagloopfilter", C.c_int), ("flagsync", C.c_int),
                 ("navbit", C.c_int)])


class ObsState(C.Structure):
    """gnsscorr_obs_t: what setobsdata() carries from call to call (ref src/sdrtrk.c:160-209)."""
    _fields_ = ([(n, C.c_double) for n in ("f_sf", "f_if", "foffset", "ctime")] + [("loopms", C.c_int)] +
                [("flagsyncf", C.c_int), ("polarity", C.c_int), ("firstsftow", C.c_double), ("firstsfcnt", C.c_uint64)] +
                [(n, C.c_double) for n in ("L", "Isum", "sumI0", "oldremcode")] +
                [("flagremcarradd", C.c_int), ("flagpolarityadd", C.c_int), ("loopcnt", C.c_uint64)])


class ObsRow(C.Structure):
    """gnsscorr_obsrow_t: element [0] of the observable histories after one call of setobsdata()."""
    _fields_ = ([(n, C.c_double) for n in ("tow", "remcout", "L", "D", "S")] + [("codei", C.c_uint64), ("cntout", C.c_uint64),
                ("snr", C.c_int), ("pad", C.c_int)])


def obs_replay(state, log_rows, II0, cnt0=0):
    """setobsdata() over one channel's log rows (numpy array of TrkLog) and its per-period trk.II[0]: returns the rows
    of observables (numpy array of ObsRow); `state` (ObsState) is updated.  Host code only."""
    L = lib()
    L.gnsscorr_obs_replay.restype = C.c_int
    L.gnsscorr_obs_replay.argtypes = [C.POINTER(ObsState), C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_int]
    log_rows = np.ascontiguousarray(log_rows)
    II0 = np.ascontiguousarray(II0, dtype=np.float64)
    n = int(log_rows.shape[0])
    out = np.zeros(max(n, 1), dtype=np.dtype(ObsRow))
    k = L.gnsscorr_obs_replay(C.byref(state), log_rows.ctypes.data, II0.ctypes.data, n, int(cnt0), out.ctypes.data, n)
    if k < 0:
        raise RuntimeError("gnsscorr_obs_replay: error %d" % k)
    return out[:k]


class FrameState(C.Structure):
    """gnsscorr_frame_t: frame synchronisation of L1 C/A on the decided nav bits (ref src/sdrnav.c:41-82)."""
    _fields_ = ([("fbits", C.c_int * 302)] + [(n, C.c_int) for n in ("polarity", "flagsyncf", "flagtow", "flagdec", "sfid", "pad")] +
                [("firstsf", C.c_uint64), ("firstsfcnt", C.c_uint64), ("firstsftow", C.c_double), ("tow_gpst", C.c_double)])


def frame_replay(state, log_rows, cnt0=0):
    """The nav bits of one channel's log rows through the preamble search / parity check / hand-over word; `state`
    (FrameState) is updated.  Host code only."""
    L = lib()
    L.gnsscorr_frame_replay.restype = C.c_int
    L.gnsscorr_frame_replay.argtypes = [C.POINTER(FrameState), C.c_void_p, C.c_int, C.c_uint64]
    log_rows = np.ascontiguousarray(log_rows)
    rc = L.gnsscorr_frame_replay(C.byref(state), log_rows.ctypes.data, int(log_rows.shape[0]), int(cnt0))
    if rc:
        raise RuntimeError("gnsscorr_frame_replay: error %d" % rc)


class AcqRes(C.Structure):
    _fields_ = [("acqcodei", C.c_int), ("freqi", C.c_int), ("acqfreq", C.c_double),
                ("cn0", C.c_double), ("peakr", C.c_double), ("flagacq", C.c_int),
                ("iters", C.c_int), ("buffloc", C.c_uint64)]


# ---- ctypes mirrors of include/sdr_compat.h (ref src/sdr.h:344-511) --------
OBSINTERPN = 80


class SdrAcq(C.Structure):
    _fields_ = [("intg", C.c_int), ("hband", C.c_double), ("step", C.c_double), ("nfreq", C.c_int),
                ("freq", C.POINTER(C.c_double)), ("acqcodei", C.c_int), ("freqi", C.c_int),
                ("acqfreq", C.c_double), ("nfft", C.c_int), ("cn0", C.c_double), ("peakr", C.c_double)]


class SdrTrkPrm(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("pllb", "dllb", "fllb", "dllw2", "dllaw", "pllw2", "pllaw", "fllw")]


class SdrTrk(C.Structure):
    _fields_ = ([(n, C.c_double) for n in ("codefreq", "carrfreq", "remcode", "remcarr", "oldremcode",
                                            "oldremcarr", "codeNco", "codeErr", "carrNco", "carrErr",
                                            "freqErr")] +
                [("buffloc", C.c_uint64), ("tow", C.c_double * OBSINTERPN),
                 ("codei", C.c_uint64 * OBSINTERPN), ("codeisum", C.c_uint64 * OBSINTERPN),
                 ("cntout", C.c_uint64 * OBSINTERPN), ("remcout", C.c_double * OBSINTERPN),
                 ("L", C.c_double * OBSINTERPN), ("D", C.c_double * OBSINTERPN),
                 ("S", C.c_double * OBSINTERPN)] +
                [(n, C.POINTER(C.c_double)) for n in ("II", "QQ", "oldI", "oldQ", "sumI", "sumQ",
                                                      "oldsumI", "oldsumQ")] +
                [("Isum", C.c_double), ("loop", C.c_int), ("loopms", C.c_int),
                 ("flagpolarityadd", C.c_int), ("flagremcarradd", C.c_int), ("flagloopfilter", C.c_int),
                 ("corrn", C.c_int), ("corrp", C.POINTER(C.c_int)), ("corrx", C.POINTER(C.c_double)),
                 ("ne", C.c_int), ("nl", C.c_int), ("prm1", SdrTrkPrm), ("prm2", SdrTrkPrm)])


class SdrCh(C.Structure):
    _fields_ = [("hsdr", C.c_ulong), ("no", C.c_int), ("sat", C.c_int), ("sys", C.c_int), ("prn", C.c_int),
                ("satstr", C.c_char * 5), ("ctype", C.c_int), ("dtype", C.c_int), ("ftype", C.c_int),
                ("f_cf", C.c_double), ("f_sf", C.c_double), ("f_if", C.c_double), ("foffset", C.c_double),
                ("code", C.POINTER(C.c_short)), ("xcode", C.c_void_p), ("clen", C.c_int),
                ("crate", C.c_double), ("ctime", C.c_double), ("ti", C.c_double), ("ci", C.c_double),
                ("nsamp", C.c_int), ("currnsamp", C.c_int), ("nsampchip", C.c_int),
                ("acq", SdrAcq), ("trk", SdrTrk), ("nav", C.c_byte * 1008),
                ("flagacq", C.c_int), ("flagtrk", C.c_int)]


class SdrIni(C.Structure):
    _fields_ = [("fend", C.c_int), ("f_cf", C.c_double * 2), ("f_sf", C.c_double * 2),
                ("f_if", C.c_double * 2), ("dtype", C.c_int * 2), ("fp1", C.c_void_p), ("fp2", C.c_void_p),
                ("file1", C.c_char * 1024), ("file2", C.c_char * 1024), ("useif1", C.c_int),
                ("useif2", C.c_int), ("nch", C.c_int), ("nchL1", C.c_int), ("nchL2", C.c_int),
                ("nchL5", C.c_int), ("nchL6", C.c_int), ("prn", C.c_int * 55), ("sys", C.c_int * 55),
                ("ctype", C.c_int * 55), ("ftype", C.c_int * 55), ("pltacq", C.c_int), ("plttrk", C.c_int),
                ("pltspec", C.c_int), ("outms", C.c_int), ("rinex", C.c_int), ("rtcm", C.c_int),
                ("sbas", C.c_int), ("log", C.c_int), ("rinexpath", C.c_char * 1024), ("rtcmport", C.c_int),
                ("sbasport", C.c_int), ("trkcorrn", C.c_int), ("trkcorrd", C.c_int), ("trkcorrp", C.c_int),
                ("trkdllb", C.c_double * 2), ("trkpllb", C.c_double * 2), ("trkfllb", C.c_double * 2),
                ("rtlsdrppmerr", C.c_int)]


class SdrStat(C.Structure):
    _fields_ = [("stopflag", C.c_int), ("specflag", C.c_int), ("buffsize", C.c_int),
                ("fendbuffsize", C.c_int), ("buff", C.c_void_p), ("buff2", C.c_void_p),
                ("tmpbuff", C.c_void_p), ("buffcnt", C.c_uint64)]


_lib = None
_KEEP = object()

# every symbol include/gnsscorr.h and include/sdr_compat.h declare and the library defines
EXPORTS_GNSSCORR = [
    "gnsscorr_last_error", "gnsscorr_device_count", "gnsscorr_create", "gnsscorr_destroy",
    "gnsscorr_stream", "gnsscorr_sync", "gnsscorr_ring_create", "gnsscorr_ring_push",
    "gnsscorr_ring_commit", "gnsscorr_ring_wrpos", "gnsscorr_ring_devptr", "gnsscorr_set_channels",
    "gnsscorr_num_channels", "gnsscorr_trk_set_state", "gnsscorr_trk_get_state", "gnsscorr_trk_run",
    "gnsscorr_trk_fetch", "gnsscorr_trk_fetch_sums", "gnsscorr_trk_devptrs", "gnsscorr_acq_run",
    "gnsscorr_acq_fetch", "gnsscorr_trk_start_from_acq", "gnsscorr_acq_power", "gnsscorr_fft16k", "gnsscorr_pspec",
    "gnsscorr_timing_enable", "gnsscorr_timing_read", "gnsscorr_timing_reset", "gnsscorr_default_ctx"]
EXPORTS_SDR = [
    "sdracquisition", "checkacquisition", "sdrtracking", "cumsumcorr", "clearcumsumcorr", "pll", "dll",
    "readinifile", "chk_initvalue", "initacqstruct", "inittrkprmstruct", "inittrkstruct", "initsdrch",
    "freesdrch", "cpxcpx", "cpxfft", "cpxifft", "cpxconv", "cpxpspec", "mixcarr", "rescode", "pcorrelator", "correlator",
    "maxvd", "meanvd", "ind2sub", "gencode", "rcvgetbuff", "file_pushtomembuf", "file_getbuff",
    "sdrnavigation", "sdrini", "sdrstat", "hbuffmtx", "hreadmtx", "hfftmtx", "hobsmtx"]


def lib():
    """The loaded libgnsscorr.so; raises GnsscorrError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnsscorrError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.gnsscorr_last_error.restype = C.c_char_p
    L.gnsscorr_stream.restype = C.c_void_p
    L.gnsscorr_stream.argtypes = [C.c_void_p]
    L.gnsscorr_ring_wrpos.restype = C.c_uint64
    L.gnsscorr_ring_wrpos.argtypes = [C.c_void_p, C.c_int]
    L.gnsscorr_ring_devptr.restype = C.c_void_p
    L.gnsscorr_ring_devptr.argtypes = [C.c_void_p, C.c_int]
    L.gnsscorr_default_ctx.restype = C.c_void_p
    L.gnsscorr_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    L.gnsscorr_destroy.argtypes = [C.c_void_p]
    L.gnsscorr_destroy.restype = None
    L.gnsscorr_sync.argtypes = [C.c_void_p]
    L.gnsscorr_ring_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p]
    L.gnsscorr_ring_push.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.gnsscorr_ring_commit.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    L.gnsscorr_ring_push_packed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.gnsscorr_ring_read.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
    L.gnsscorr_set_channels.argtypes = [C.c_void_p, C.c_int, C.POINTER(ChanDesc)]
    L.gnsscorr_num_channels.argtypes = [C.c_void_p]
    L.gnsscorr_trk_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(TrkState)]
    L.gnsscorr_trk_get_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(TrkState)]
    L.gnsscorr_trk_run.argtypes = [C.c_void_p, C.c_int]
    L.gnsscorr_trk_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gnsscorr_trk_fetch_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.gnsscorr_trk_devptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.gnsscorr_loop_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(LoopState)]
    L.gnsscorr_loop_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(LoopState)]
    L.gnsscorr_trk_run_loop.argtypes = [C.c_void_p, C.c_int]
    L.gnsscorr_trk_fetch_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.gnsscorr_acq_run.argtypes = [C.c_void_p, C.c_uint64]
    L.gnsscorr_acq_fetch.argtypes = [C.c_void_p, C.POINTER(AcqRes)]
    L.gnsscorr_trk_start_from_acq.argtypes = [C.c_void_p]
    L.gnsscorr_acq_power.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.gnsscorr_fft16k.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.gnsscorr_pspec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.gnsscorr_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.gnsscorr_timing_reset.argtypes = [C.c_void_p]
    L.gnsscorr_timing_read.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    # reference-named symbols (bound when present; tests/test_abi.py checks that all are)
    def _sig(name, restype, argtypes):
        try:
            f = getattr(L, name)
        except AttributeError:
            return
        if restype is not _KEEP:
            f.restype = restype
        f.argtypes = argtypes

    _sig("gencode", C.POINTER(C.c_short), [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)])
    _sig("initsdrch", _KEEP, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(SdrCh)])
    _sig("freesdrch", None, [C.POINTER(SdrCh)])
    _sig("sdrtracking", C.c_uint64, [C.POINTER(SdrCh), C.c_uint64, C.c_uint64])
    _sig("sdracquisition", C.c_uint64, [C.POINTER(SdrCh), C.c_void_p])
    _sig("checkacquisition", _KEEP, [C.c_void_p, C.POINTER(SdrCh)])
    _sig("correlator", None, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_int])
    _sig("pcorrelator", None, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p])
    _sig("cumsumcorr", None, [C.POINTER(SdrTrk), C.c_int])
    _sig("clearcumsumcorr", None, [C.POINTER(SdrTrk)])
    _sig("pll", None, [C.POINTER(SdrCh), C.POINTER(SdrTrkPrm), C.c_double])
    _sig("dll", None, [C.POINTER(SdrCh), C.POINTER(SdrTrkPrm), C.c_double])
    _sig("readinifile_at", _KEEP, [C.POINTER(SdrIni), C.c_char_p])
    _sig("rcvinit_file", _KEEP, [C.POINTER(SdrIni)])
    _sig("file_pushtomembuf", None, [])
    _sig("chk_initvalue", _KEEP, [C.POINTER(SdrIni)])
    _sig("cpxpspec", None, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p])
    _sig("cpxfft", None, [C.c_void_p, C.c_void_p, C.c_int])
    _sig("cpxifft", None, [C.c_void_p, C.c_void_p, C.c_int])
    _sig("cpxcpx", None, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p])
    _sig("cpxconv", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p])
    _sig("mixcarr", C.c_double, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p])
    _sig("rescode", C.c_double, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p])
    _sig("maxvd", C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)])
    _sig("meanvd", C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int])
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise GnsscorrError(lib().gnsscorr_last_error().decode())


def sdrini():
    return SdrIni.in_dll(lib(), "sdrini")


def sdrstat():
    return SdrStat.in_dll(lib(), "sdrstat")


def gencode(prn, ctype):
    """gencode() of the library (ref src/sdrcode.c:523-539): (int16 chips, chip rate)."""
    n, cr = C.c_int(), C.c_double()
    p = lib().gencode(prn, ctype, C.byref(n), C.byref(cr))
    if not p:
        raise GnsscorrError(f"gencode failed for prn {prn} ctype {ctype}")
    code = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    C.CDLL(None).free(p)
    return code, cr.value


class Channel:
    """Constants of one receiver channel, derived exactly as initsdrch() does
    (ref src/sdrinit.c:583-657, acquisition grid :385-394,:633-635, taps :446-455)."""

    def __init__(self, prn, ctype=CTYPE_L1CA, dtype=DTYPEIQ, ftype=FTYPE1, f_cf=1575.42e6,
                 f_sf=16.368e6, f_if=0.0, corrn=2, corrd=3, corrp=3, hband=7000, step=200, intg=10):
        self.prn, self.ctype, self.dtype, self.ftype = prn, ctype, dtype, ftype
        self.code, self.crate = gencode(prn, ctype)
        self.clen = len(self.code)
        self.f_sf, self.f_if = f_sf, f_if
        self.ti = 1 / f_sf
        self.ci = self.ti * self.crate
        self.ctime = self.clen / self.crate
        self.nsamp = int(f_sf * self.ctime)
        self.nsampchip = int(self.nsamp / self.clen)
        if ctype == CTYPE_G1:
            self.f_cf = 1.60200e9 + 0.56250e6 * prn
            self.foffset = 0.56250e6 * prn
        else:
            self.f_cf, self.foffset = f_cf, 0.0
        self.intg = intg
        self.nfreq = 2 * (hband // step) + 1
        self.nfft = 2 * self.nsamp
        self.freq = np.array([f_if + ((i - (self.nfreq - 1) // 2) * float(step)) + self.foffset
                              for i in range(self.nfreq)], dtype=np.float64)
        self.corrn = corrn
        self.corrp = np.array([corrd * (i + 1) for i in range(corrn)], dtype=np.int32)
        self.ne = self.nl = 0
        for i in range(corrn):
            if self.corrp[i] == corrp:
                self.ne, self.nl = 2 * (i + 1) - 1, 2 * (i + 1)
        self.ntap = 1 + 2 * corrn

    def desc(self):
        d = ChanDesc()
        d.prn, d.ctype, d.dtype, d.ftype = self.prn, self.ctype, self.dtype, self.ftype
        d.clen, d.nsamp, d.nsampchip = self.clen, self.nsamp, self.nsampchip
        d.f_sf, d.f_if, d.foffset = self.f_sf, self.f_if, self.foffset
        d.crate, d.ctime, d.ti = self.crate, self.ctime, self.ti
        self._code16 = np.ascontiguousarray(self.code, dtype=np.int16)
        d.code = self._code16.ctypes.data_as(C.POINTER(C.c_short))
        d.intg, d.nfreq, d.nfft = self.intg, self.nfreq, self.nfft
        d.freq = self.freq.ctypes.data_as(C.POINTER(C.c_double))
        d.corrn = self.corrn
        d.corrp = self.corrp.ctypes.data_as(C.POINTER(C.c_int))
        return d


class Engine:
    """One GPU context of libgnsscorr (see include/gnsscorr.h for the semantics)."""

    def __init__(self, device=0, stream=None):
        self._L = lib()
        h = C.c_void_p()
        _check(self._L.gnsscorr_create(C.byref(h), device, stream))
        self.h = h
        self.channels = []

    def close(self):
        if self.h:
            self._L.gnsscorr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._L.gnsscorr_stream(self.h)

    def stream_ptr(self):
        """The context's HIP stream as an integer (for torch.cuda.ExternalStream)."""
        return int(self._L.gnsscorr_stream(self.h) or 0)

    def sync(self):
        _check(self._L.gnsscorr_sync(self.h))

    # -- ring
    def ring_create(self, ftype, dtype, ringlen, devptr=None):
        _check(self._L.gnsscorr_ring_create(self.h, ftype, dtype, ringlen, devptr))

    def ring_push(self, ftype, samples):
        a = np.ascontiguousarray(samples, dtype=np.int8)
        dtype = 2 if (a.ndim == 2 and a.shape[1] == 2) else None
        n = a.shape[0] if dtype == 2 else a.size
        _check(self._L.gnsscorr_ring_push(self.h, ftype, a.ctypes.data, n))

    def ring_push_raw(self, ftype, raw_bytes, nsamp):
        a = np.ascontiguousarray(raw_bytes, dtype=np.int8)
        _check(self._L.gnsscorr_ring_push(self.h, ftype, a.ctypes.data, nsamp))

    def ring_push_packed(self, fmt, packed, nsamp):
        """fmt: FMT_STEREO (one byte per sample instant) or FMT_RTLSDR (2*nsamp unsigned bytes)"""
        a = np.ascontiguousarray(packed, dtype=np.uint8)
        _check(self._L.gnsscorr_ring_push_packed(self.h, fmt, a.ctypes.data, nsamp))

    def ring_read(self, ftype, buffloc, n, dtype):
        out = np.empty((n, 2) if dtype == 2 else (n,), np.int8)
        _check(self._L.gnsscorr_ring_read(self.h, ftype, buffloc, n, out.ctypes.data))
        return out

    def ring_commit(self, ftype, nsamp):
        _check(self._L.gnsscorr_ring_commit(self.h, ftype, nsamp))

    def ring_wrpos(self, ftype):
        return self._L.gnsscorr_ring_wrpos(self.h, ftype)

    def ring_devptr(self, ftype):
        return self._L.gnsscorr_ring_devptr(self.h, ftype)

    # -- channels
    def set_channels(self, channels):
        arr = (ChanDesc * len(channels))(*[c.desc() for c in channels])
        _check(self._L.gnsscorr_set_channels(self.h, len(channels), arr))
        self.channels = list(channels)

    # -- tracking
    def trk_set_state(self, states, ch0=0):
        arr = (TrkState * len(states))()
        for i, s in enumerate(states):
            arr[i].carrfreq, arr[i].codefreq = s["carrfreq"], s["codefreq"]
            arr[i].remcode, arr[i].remcarr, arr[i].buffloc = s["remcode"], s["remcarr"], s["buffloc"]
        _check(self._L.gnsscorr_trk_set_state(self.h, ch0, len(states), arr))

    def trk_get_state(self, ch0=0, nch=None):
        nch = len(self.channels) - ch0 if nch is None else nch
        arr = (TrkState * nch)()
        _check(self._L.gnsscorr_trk_get_state(self.h, ch0, nch, arr))
        return [dict(carrfreq=a.carrfreq, codefreq=a.codefreq, remcode=a.remcode, remcarr=a.remcarr,
                     buffloc=a.buffloc) for a in arr]

    def trk_run(self, nepoch):
        _check(self._L.gnsscorr_trk_run(self.h, nepoch))
        self._nepoch = nepoch

    def trk_fetch(self):
        nch, ne, nt = len(self.channels), self._nepoch, self.channels[0].ntap
        II = np.empty((nch, ne, nt), np.float64)
        QQ = np.empty((nch, ne, nt), np.float64)
        ns = np.empty((nch, ne), np.int32)
        _check(self._L.gnsscorr_trk_fetch(self.h, II.ctypes.data, QQ.ctypes.data, ns.ctypes.data))
        return II, QQ, ns

    def trk_fetch_sums(self):
        nch, nt = len(self.channels), self.channels[0].ntap
        sI = np.empty((nch, nt), np.float64)
        sQ = np.empty((nch, nt), np.float64)
        _check(self._L.gnsscorr_trk_fetch_sums(self.h, sI.ctypes.data, sQ.ctypes.data))
        return sI, sQ

    # -- closed loop
    def loop_state(self, ch, acqfreq, dllb=(5.0, 1.0), pllb=(30.0, 10.0), fllb=(200.0, 50.0), flagsync=0, synci=0,
                   cnt=0, loop=None):
        """A LoopState for channel index `ch` with the constants inittrkprmstruct()/inittrkstruct() derive
        (ref src/sdrinit.c:402-425,432-480) and zeroed filter state."""
        c = self.channels[ch]
        ls = LoopState()
        ls.acqfreq, ls.f_if, ls.foffset, ls.f_cf, ls.crate, ls.ctime = acqfreq, c.f_if, c.foffset, c.f_cf, c.crate, c.ctime
        for i in range(2):
            ls.dllw2[i] = (dllb[i] / 0.53) * (dllb[i] / 0.53)
            ls.dllaw[i] = 1.414 * (dllb[i] / 0.53)
            ls.pllw2[i] = (pllb[i] / 0.53) * (pllb[i] / 0.53)
            ls.pllaw[i] = 1.414 * (pllb[i] / 0.53)
            ls.fllw[i] = fllb[i] / 0.25
        ls.ne, ls.nl = c.ne, c.nl
        loop = loop if loop is not None else (2 if c.ctype == CTYPE_L1SBAS else 10)
        ls.loopms = loop * int(c.ctime * 1000)
        ls.rate = 10 if c.ctype == CTYPE_G1 else (2 if c.ctype == CTYPE_L1SBAS else 20)
        ls.flagsync, ls.synci, ls.cnt = flagsync, synci, cnt
        ls.prn = c.prn
        return ls

    def loop_set(self, states, ch0=0):
        arr = (LoopState * len(states))(*states)
        _check(self._L.gnsscorr_loop_set(self.h, ch0, len(states), arr))

    def loop_get(self, ch0=0, nch=None):
        nch = len(self.channels) - ch0 if nch is None else nch
        arr = (LoopState * nch)()
        _check(self._L.gnsscorr_loop_get(self.h, ch0, nch, arr))
        return list(arr)

    def trk_run_loop(self, nperiod):
        _check(self._L.gnsscorr_trk_run_loop(self.h, nperiod))
        self._nepoch = nperiod

    def trk_fetch_log(self):
        nch = len(self.channels)
        log = np.zeros((nch, self._nepoch), dtype=np.dtype(TrkLog))
        ndone = np.zeros(nch, np.int32)
        _check(self._L.gnsscorr_trk_fetch_log(self.h, log.ctypes.data, ndone.ctypes.data))
        return log, ndone

    # -- acquisition
    def acq_run(self, wrpos=0):
        _check(self._L.gnsscorr_acq_run(self.h, wrpos))

    def acq_fetch(self):
        arr = (AcqRes * len(self.channels))()
        _check(self._L.gnsscorr_acq_fetch(self.h, arr))
        return [dict(acqcodei=a.acqcodei, freqi=a.freqi, acqfreq=a.acqfreq, cn0=a.cn0, peakr=a.peakr,
                     flagacq=a.flagacq, iters=a.iters, buffloc=a.buffloc) for a in arr]

    def trk_start_from_acq(self):
        _check(self._L.gnsscorr_trk_start_from_acq(self.h))

    def acq_power(self, ch):
        c = self.channels[ch]
        P = np.empty((c.nfreq, c.nsamp), np.float64)
        _check(self._L.gnsscorr_acq_power(self.h, ch, P.ctypes.data))
        return P

    # -- timing
    def timing(self, on=True):
        _check(self._L.gnsscorr_timing_enable(self.h, int(on)))

    def timing_reset(self):
        _check(self._L.gnsscorr_timing_reset(self.h))

    def timing_read(self, kernel):
        ms, n = C.c_double(), C.c_int()
        _check(self._L.gnsscorr_timing_read(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
