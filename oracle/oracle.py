"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see
gnss_oracle.h).  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
MAXTAPS = 33


class AcqRes(C.Structure):
    _fields_ = [("acqcodei", C.c_int), ("freqi", C.c_int), ("acqfreq", C.c_double), ("cn0", C.c_double),
                ("peakr", C.c_double), ("acquired", C.c_int)]


class Ring(C.Structure):
    _fields_ = [("buff", C.c_void_p), ("ringlen", C.c_uint64), ("wrpos", C.c_uint64)]


class Chan(C.Structure):
    _fields_ = ([("dtype", C.c_int), ("clen", C.c_int), ("nsamp", C.c_int), ("nsampchip", C.c_int)] +
                [(n, C.c_double) for n in ("f_sf", "f_if", "foffset", "f_cf", "crate", "ctime", "ti", "ci")] +
                [("code", C.c_short * 1023), ("intg", C.c_int), ("nfreq", C.c_int), ("nfft", C.c_int),
                 ("freq", C.c_double * 256), ("xcode", C.c_void_p), ("acq", AcqRes), ("flagacq", C.c_int),
                 ("flagtrk", C.c_int), ("corrn", C.c_int), ("corrp", C.c_int * 16), ("ne", C.c_int),
                 ("nl", C.c_int), ("loopms", C.c_int)] +
                [(n, C.c_double) for n in ("codefreq", "carrfreq", "remcode", "remcarr", "oldremcode",
                                            "oldremcarr", "codeNco", "codeErr", "carrNco", "carrErr",
                                            "freqErr")] +
                [(n, C.c_double * MAXTAPS) for n in ("II", "QQ", "oldI", "oldQ", "sumI", "sumQ", "oldsumI",
                                                     "oldsumQ")] +
                [("currnsamp", C.c_int)] +
                [(n, C.c_double * 2) for n in ("dllw2", "dllaw", "pllw2", "pllaw", "fllw")] +
                [(n, C.c_int) for n in ("rate", "flagsync", "synci", "navcnt", "swloop", "flagloopfilter")] +
                [("cnt", C.c_uint64)] +
                [(n, C.c_int) for n in ("prn", "biti", "bit", "swsync", "swreset", "flagpol")] +
                [("bitIP", C.c_double), ("bitsync", C.c_int * 20)] +
                [(n, C.c_double) for n in ("obs_tow", "obs_remcout", "obs_L", "obs_D", "obs_S", "obs_Isum")] +
                [(n, C.c_uint64) for n in ("obs_codei", "obs_cntout", "obs_codeisum", "firstsfcnt", "loopcnt")] +
                [("firstsftow", C.c_double)] +
                [(n, C.c_int) for n in ("flagremcarradd", "flagpolarityadd", "flagsyncf", "polarity", "obs_n", "obs_nsnr")])


class Frame(C.Structure):
    _fields_ = [("fbits", C.c_int * 302), ("fbitsdec", C.c_int * 302)] + [(n, C.c_int) for n in (
        "polarity", "flagsyncf", "flagtow", "flagdec", "sfid")] + [("firstsf", C.c_uint64), ("firstsfcnt", C.c_uint64),
        ("firstsftow", C.c_double), ("tow_gpst", C.c_double)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    d, i, vp = C.c_double, C.c_int, C.c_void_p
    L.orc_gencode.argtypes = [i, i, vp, C.POINTER(i), C.POINTER(d)]
    L.orc_carrier_lut.argtypes = [vp, vp]
    L.orc_mixcarr_seq.restype = d
    L.orc_mixcarr_seq.argtypes = [vp, i, d, i, d, d, vp, vp]
    L.orc_rescode_seq.restype = d
    L.orc_rescode_seq.argtypes = [vp, i, d, i, d, i, vp]
    L.orc_correlator.restype = None
    L.orc_correlator.argtypes = [vp, i, d, i, d, d, d, d, vp, i, vp, vp, C.POINTER(d), C.POINTER(d), vp, i]
    L.orc_cpxcpx.argtypes = [vp, vp, d, i, vp]
    L.orc_fft.argtypes = [vp, i, i]
    L.orc_cpxconv.argtypes = [vp, vp, i, i, i, vp]
    L.orc_cpxpspec.argtypes = [vp, i, i, vp]
    L.orc_codespectrum.argtypes = [vp, i, d, i, i, vp]
    L.orc_pcorrelator.argtypes = [vp, i, d, i, vp, i, d, i, vp, vp]
    L.orc_pcorrelator_td.argtypes = [vp, i, d, i, vp, i, i, vp, i, d, i, i, vp]
    L.orc_maxvd.restype = d
    L.orc_maxvd.argtypes = [vp, i, i, i, C.POINTER(i)]
    L.orc_meanvd.restype = d
    L.orc_meanvd.argtypes = [vp, i, i, i]
    L.orc_checkacquisition.argtypes = [vp, i, i, i, d, vp, C.POINTER(AcqRes)]
    L.orc_getbuff.argtypes = [C.POINTER(Ring), C.c_uint64, i, i, vp]
    L.orc_initchan.argtypes = [C.POINTER(Chan), i, i, i, d, d, d, i, i, i, vp, vp, vp]
    L.orc_sdracquisition.restype = C.c_uint64
    L.orc_sdracquisition.argtypes = [C.POINTER(Chan), C.POINTER(Ring), vp, C.POINTER(i)]
    L.orc_sdrtracking.restype = C.c_uint64
    L.orc_sdrtracking.argtypes = [C.POINTER(Chan), C.POINTER(Ring), C.c_uint64]
    L.orc_cumsumcorr.argtypes = [C.POINTER(Chan), i]
    L.orc_clearcumsumcorr.argtypes = [C.POINTER(Chan)]
    L.orc_pll.argtypes = [C.POINTER(Chan), i, d]
    L.orc_dll.argtypes = [C.POINTER(Chan), i, d]
    L.orc_sdrthread_step.argtypes = [C.POINTER(Chan), C.POINTER(Ring), C.POINTER(C.c_uint64)]
    L.orc_setobsdata.argtypes = [C.POINTER(Chan), C.c_uint64, C.c_uint64, i]
    L.orc_setobsdata.restype = None
    L.orc_navframe_l1ca.argtypes = [C.POINTER(Frame), i, C.c_uint64, C.c_uint64]
    L.orc_navframe_l1ca.restype = None
    L.orc_stereo_exp.argtypes = [vp, i, i, vp]
    L.orc_rtlsdr_exp.argtypes = [vp, i, vp]
    _lib = L
    return L


def gencode(prn, ctype):
    code = np.zeros(1023, np.int16)
    n, cr = C.c_int(), C.c_double()
    if lib().orc_gencode(prn, ctype, code.ctypes.data, C.byref(n), C.byref(cr)) < 0:
        raise ValueError("orc_gencode")
    return code[:n.value].copy(), cr.value


def correlator(data, dtype, ti, n, freq, phi0, crate, coff, s, code):
    """-> (II, QQ, remc, remp) as the reference's correlator() returns them."""
    data = np.ascontiguousarray(data, np.int8)
    s = np.ascontiguousarray(s, np.int32)
    code = np.ascontiguousarray(code, np.int16)
    nt = 1 + 2 * len(s)
    II, QQ = np.zeros(nt), np.zeros(nt)
    remc, remp = C.c_double(), C.c_double()
    lib().orc_correlator(data.ctypes.data, dtype, ti, n, freq, phi0, crate, coff, s.ctypes.data, len(s),
                         II.ctypes.data, QQ.ctypes.data, C.byref(remc), C.byref(remp), code.ctypes.data,
                         len(code))
    return II, QQ, remc.value, remp.value


def make_chan(prn, ctype=1, dtype=2, f_cf=1575.42e6, f_sf=16.368e6, f_if=0.0, corrn=2, corrd=3, corrp=3,
              dllb=(5.0, 1.0), pllb=(30.0, 10.0), fllb=(200.0, 50.0)):
    ch = Chan()
    a = [np.array(x, np.float64) for x in (dllb, pllb, fllb)]
    if lib().orc_initchan(C.byref(ch), prn, ctype, dtype, f_cf, f_sf, f_if, corrn, corrd, corrp,
                          a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data) < 0:
        raise ValueError("orc_initchan")
    return ch


def make_ring(buff, ringlen, wrpos):
    r = Ring()
    r.buff = buff.ctypes.data
    r.ringlen = ringlen
    r.wrpos = wrpos
    return r


def codespectrum(ch):
    x = np.zeros(2 * ch.nfft, np.float32)
    code = np.ctypeslib.as_array(ch.code)
    lib().orc_codespectrum(code.ctypes.data, ch.clen, ch.ci, ch.nsamp, ch.nfft, x.ctypes.data)
    return x
