import ctypes as C, numpy as np, sys
sys.path.insert(0, '/root/repo')
lib = C.CDLL('/tmp/libhp_asan.so')
class Cfg(C.Structure):
    _fields_ = [("centre_frequency", C.c_double), ("bandwidth", C.c_double), ("dispersion_measure", C.c_double), ("input_nchan", C.c_uint32),
                ("nchan", C.c_uint32), ("ndim", C.c_uint32), ("dual_sideband", C.c_int32), ("dc_centred", C.c_uint32), ("swap", C.c_uint32),
                ("freq_res", C.c_uint32), ("ndat_max", C.c_uint32), ("fractional_delay", C.c_uint32)]
class Info(C.Structure):
    _fields_ = [("impulse_pos", C.c_uint32), ("impulse_neg", C.c_uint32), ("minimum_ndat", C.c_uint32), ("ndat", C.c_uint32)]
lib.dspsr_amd_optimal_fft_length.restype = C.c_uint64
lib.dspsr_amd_optimal_fft_length.argtypes = [C.c_uint64, C.c_uint64]
lib.dspsr_amd_eight_bit_scale.restype = C.c_double
lib.dspsr_amd_eight_bit_scale.argtypes = [C.c_double]
n = 0
for f0, bw, dm, nchan, x, innch, ndim, fd in [(1382, -400, 1000, 1024, 4096, 1, 1, 0), (1382, -400, 67.99, 64, 0, 1, 1, 1), (1400, 64, 10, 16, 0, 1, 2, 0),
                                              (1382, -50, 1000, 512, 0, 1, 2, 1), (1382, -400, 500, 256, 4096, 8, 2, 0), (2000, -400, 500, 256, 4096, 1, 1, 0)]:
    cfg = Cfg(f0, bw, dm, innch, nchan, ndim, -1, 0, 0, x, 0, fd)
    info = Info()
    err = C.create_string_buffer(256)
    rc = lib.dspsr_amd_dedispersion_prepare(C.byref(cfg), C.byref(info), err, 256)
    if rc == 0:
        k = np.zeros(2 * nchan * info.ndat, np.float32)
        rc2 = lib.dspsr_amd_dedispersion_build(C.byref(cfg), info.ndat, k.ctypes.data_as(C.c_void_p))
        assert rc2 == 0 and np.isfinite(k).all()
        n += 1
    else:
        print("prepare rc", rc, err.value.decode())
for nbad in (1, 3, 54, 844, 1687, 6735, 14567, 100000):
    lib.dspsr_amd_optimal_fft_length(nbad, 0); lib.dspsr_amd_optimal_fft_length(nbad, 1 << 16)
plan = np.zeros(100000, np.uint32); hits = np.zeros(1024, np.uint32)
lib.dspsr_amd_fold_binplan.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
assert lib.dspsr_amd_fold_binplan(0.731, 1.0 / 345.67, 1024, 100000, plan.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p)) == 0
d = np.zeros(1024, np.int64)
lib.dspsr_amd_dedispersion_sample_delays.argtypes = [C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
assert lib.dspsr_amd_dedispersion_sample_delays(1382.0, -400.0, 71.0, 1024, 390625.0, 0, 0, 0, d.ctypes.data_as(C.c_void_p)) == 0
assert lib.dspsr_amd_dedispersion_sample_delays(1382.0, -400.0, 71.0, 256, 390625.0, 1, 4, 1, d.ctypes.data_as(C.c_void_p)) == 0
print("host_prep under ASan/UBSan: %d kernels built, no report" % n, lib.dspsr_amd_eight_bit_scale(0.02957))
