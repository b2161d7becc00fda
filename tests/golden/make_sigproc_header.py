"""Generates tests/golden/sigproc_header.bin with the REFERENCE's own header writer (oracle/_ref/libsigproc_ref.so,
built by oracle/Makefile from Kernel/Formats/sigproc/*.c where they lie under /root/reference): the globals are set
the way dsp::SigProcObservation::unload does (SigProcObservation.C:228-270) and filterbank_header() writes the file.
Run in the build container; the fixture and sigproc_header.json (the values) are committed, the reference is not."""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VALUES = dict(source_name="J0835-4510", rawdatafile="unknown", machine_id=0, telescope_id=0, src_raj=83520.61, src_dej=-451034.8,
              fch1=1581.951171875, foff=-0.09765625, nchans=4096, nbits=8, tstart_mjd=55299.087326388886, tsamp=0.00016384, nifs=1)


def reference_header_bytes(values, path):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsigproc_ref.so"))
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    def setv(name, ctype, v):
        ctype.in_dll(lib, name).value = v

    def sets(name, n, v):
        buf = (C.c_char * n).in_dll(lib, name)
        buf.value = v.encode("ascii")
    sets("source_name", 80, values["source_name"])
    sets("inpfile", 80, values["rawdatafile"])
    setv("machine_id", C.c_int, values["machine_id"])
    setv("telescope_id", C.c_int, values["telescope_id"])
    setv("src_raj", C.c_double, values["src_raj"])
    setv("src_dej", C.c_double, values["src_dej"])
    setv("az_start", C.c_double, 0.0)
    setv("za_start", C.c_double, 0.0)
    setv("fch1", C.c_double, values["fch1"])
    setv("foff", C.c_double, values["foff"])
    setv("nchans", C.c_int, values["nchans"])
    setv("nbits", C.c_int, values["nbits"])
    setv("obits", C.c_int, values["nbits"])
    setv("tstart", C.c_double, values["tstart_mjd"])
    setv("tsamp", C.c_double, values["tsamp"])
    setv("nifs", C.c_int, values["nifs"])
    setv("nbeams", C.c_int, 0)
    setv("ibeam", C.c_int, 0)
    setv("sumifs", C.c_int, 0)
    setv("headerless", C.c_int, 0)
    setv("zerolagdump", C.c_int, 0)
    setv("swapout", C.c_int, 0)
    C.c_float.in_dll(lib, "start_time").value = 0.0
    ifs = (C.c_char * 8).in_dll(lib, "ifstream")
    for i in range(values["nifs"]):
        ifs[i] = b"Y"
    f = libc.fopen(path.encode(), b"wb")
    lib.filterbank_header.argtypes = [C.c_void_p]
    lib.filterbank_header(f)
    libc.fclose(f)
    return open(path, "rb").read()


if __name__ == "__main__":
    b = reference_header_bytes(VALUES, os.path.join(HERE, "sigproc_header.bin"))
    json.dump(VALUES, open(os.path.join(HERE, "sigproc_header.json"), "w"), indent=1)
    print("wrote %d bytes" % len(b))
