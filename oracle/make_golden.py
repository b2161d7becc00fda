"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference exists):  python -m oracle.make_golden

  ref_kat.npz       outputs of the reference's OWN C files (oracle/_ref/libdspsr_ref.so =
                    cross_detect.c, stokes_detect.c, optimize_fft.c built unmodified) on seeded inputs
  chirp_kat.npz     first/last 8 phasors per channel + impulse_pos/neg for 3 configurations (oracle)
  e2e_small.npz     tiny end-to-end case: raw bytes, hits, folded sums in float64 and float32 (oracle)
  vela_polyco.json  the numeric content of the reference's Benchmark/vela.polyco (a data file)
"""
import ctypes as C
import json
import os

import numpy as np

from dspsr_amd import synth
from oracle import dspsr_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdspsr_ref.so"))
    rng = np.random.default_rng(20100413)
    ndat = 64
    p = rng.standard_normal(2 * ndat).astype(np.float32)
    q = rng.standard_normal(2 * ndat).astype(np.float32)
    cross = np.zeros((4, ndat), np.float32)
    stokes = np.zeros((4, ndat), np.float32)
    ref.cross_detect(ndat, fp(p), fp(q), fp(cross[0]), fp(cross[1]), fp(cross[2]), fp(cross[3]), 1)
    ref.stokes_detect(ndat, fp(p), fp(q), fp(stokes[0]), fp(stokes[1]), fp(stokes[2]), fp(stokes[3]), 1)
    ref.optimal_fft_length.restype = C.c_uint64
    ref.optimal_fft_length.argtypes = [C.c_uint64, C.c_uint64, C.c_char]
    nbad = np.array([54, 844, 1687, 1909, 6735, 14567], np.uint64)
    nopt = np.array([ref.optimal_fft_length(int(n), 0, b"\0") for n in nbad], np.uint64)
    np.savez(os.path.join(GOLD, "ref_kat.npz"), p=p, q=q, cross=cross, stokes=stokes, nbad=nbad, nopt=nopt)

    cfgs = np.array([[1382.0, -400.0, 1000.0, 1024, 4096], [2000.0, -400.0, 500.0, 256, 4096],
                     [1400.0, 64.0, 10.0, 16, 1024]])
    out = {"cfg": cfgs}
    for i, (f0, bw, dm, nchan, nd) in enumerate(cfgs):
        obs = o.Observation(centre_frequency=f0, bandwidth=bw, dispersion_measure=dm)
        d = o.Dedispersion()
        d.set_frequency_resolution(int(nd))
        d.match(obs, int(nchan))
        k = d.buffer.reshape(int(nchan), int(nd))
        out["phasors_%d" % i] = np.concatenate([k[:, :8], k[:, -8:]], axis=1)
        out["impulse_%d" % i] = np.array([d.impulse_pos, d.impulse_neg])
    np.savez(os.path.join(GOLD, "chirp_kat.npz"), **out)

    prm = dict(freq=1382.0, bw=-8.0, tsamp_us=1.0 / 16.0, dm=20.0, period=0.002, nchan=8, nbin=64, freq_res=512)
    obs = o.Observation(centre_frequency=prm["freq"], bandwidth=prm["bw"], tsamp_us=prm["tsamp_us"],
                        dispersion_measure=prm["dm"])
    resp = o.Dedispersion()
    resp.set_frequency_resolution(prm["freq_res"])
    resp.match(obs, prm["nchan"])
    plan = o.filterbank_plan(obs, prm["nchan"], resp)
    prm["ndat"] = int(4 * plan.nsamp_step + plan.nsamp_overlap)
    raw = synth.voltages(prm["ndat"], prm["freq"], prm["bw"], prm["tsamp_us"], prm["dm"], prm["period"])
    res = {}
    for dtype, key in ((np.float64, "profile64"), (np.float32, "profile32")):
        ps, *_ = o.run_pipeline(raw, obs, prm["nchan"], prm["dm"], prm["nbin"], folding_period=prm["period"],
                                freq_res=prm["freq_res"], dtype=dtype)
        res[key] = ps.data
        res["hits"] = ps.hits
    np.savez_compressed(os.path.join(GOLD, "e2e_small.npz"), raw=raw, params=json.dumps(prm), **res)

    pc = "/root/reference/Benchmark/vela.polyco"
    if os.path.exists(pc):
        json.dump({"text": open(pc).read()}, open(os.path.join(GOLD, "vela_polyco.json"), "w"))
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
