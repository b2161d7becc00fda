#!/usr/bin/env python3
"""tools/dspsr_amd_fold.py : the path behind the reference's own option spellings (dspsr.C:207-510) -- a thin driver over
dspsr_amd.dada.fold_file for trying the engine on a DADA file; not a re-implementation of the dspsr application.

  dspsr_amd_fold.py -F 1024:D -D 1000 -b 1024 -c 0.0893 [-x 4096] [-L 10 | -s | -turns N] [-P polyco] [-K] [-d 4] [-r]
  dspsr_amd_fold.py -F 128 ...   (no `:D`: filterbank, THEN coherent dedispersion per channel -- Filterbank::Config::After; -D 0: none)
  dspsr_amd_fold.py -F 128:B ... (coherent dedispersion of the whole band, THEN the filterbank -- Filterbank::Config::Before)
                    [--dump Detection] [--dump Fold] [-O out_prefix] file.dada

Every completed sub-integration is written as <prefix>_<n>.ps (the PhaseSeries hand-off file of INTEGRATION.md:
raw sums + hits; dsp::Archiver's normalisation is the reader's job)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("file")
    ap.add_argument("-F", dest="fb", required=True, help="nchan:D (convolving filterbank, coherent dedispersion During) or nchan (filterbank, then dsp::Convolution: After)")
    ap.add_argument("-D", dest="dm", type=float, default=None, help="dispersion measure (default: DM of the header)")
    ap.add_argument("-b", dest="nbin", type=int, default=0, help="phase bins (default: dsp::Fold::choose_nbin)")
    ap.add_argument("-c", dest="period", type=float, default=0.0, help="constant folding period in seconds")
    ap.add_argument("-P", dest="polyco", default=None, help="TEMPO polyco file")
    ap.add_argument("-x", dest="nfft", type=int, default=0, help="response (FFT) length per channel")
    ap.add_argument("-L", dest="subint", type=float, default=0.0, help="sub-integration length in seconds")
    ap.add_argument("-s", dest="single", action="store_true", help="single pulses (one turn per sub-integration)")
    ap.add_argument("-turns", dest="turns", type=float, default=0.0, help="turns per sub-integration")
    ap.add_argument("-K", dest="interchan", action="store_true", help="remove the inter-channel dispersion delay")
    ap.add_argument("-d", dest="ndim", type=int, default=4, choices=[1, 2, 4], help="detected layout (ndim)")
    ap.add_argument("-r", dest="record", action="store_true", help="report the time spent in each operation")
    ap.add_argument("--dump", action="append", default=[], help="dump the input of this operation (Detection, Fold)")
    ap.add_argument("-O", dest="prefix", default="dspsr_amd", help="output file name prefix")
    ap.add_argument("--cuda", dest="device", type=int, default=0, help="device id (the reference's spelling)")
    a = ap.parse_args()
    when = "during" if a.fb.endswith(":D") else "before" if a.fb.endswith(":B") else "after"
    if ":" in a.fb and when == "after":
        sys.exit("-F nchan:D (coherent dedispersion During the filterbank), -F nchan (After it) or -F nchan:B (Before it) are on this "
                 "path; not %s" % a.fb)
    import torch
    from dspsr_amd import dada, pipeline
    hdr, _ = dada.read_header(a.file)
    info, extras = dada.observation(hdr)
    dm = a.dm if a.dm is not None else extras["dm"]
    if dm is None:
        sys.exit("no -D and no DM in the header")
    polyco = pipeline.Polyco(open(a.polyco).read()) if a.polyco else None
    if polyco is None and a.period <= 0:
        sys.exit("dsp::Fold::fold no polynomial and no period specified (-c or -P)")
    nchan = int(a.fb.split(":")[0])
    pfold = a.period if a.period > 0 else 1.0 / polyco.frequency(info.mjd_day, info.mjd_sec)
    out_rate = info.rate / (2 if info.ndim == 1 else 1) / (nchan // info.nchan)
    nbin = a.nbin or pipeline.choose_nbin(pfold, out_rate)
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=a.period, freq_res=a.nfft,
                          subint_seconds=a.subint, subint_turns=1.0 if a.single else a.turns, ndim=a.ndim,
                          interchan_dedispersion=a.interchan, record_time=a.record,
                          convolve_when="never" if when == "after" and dm == 0.0 else when)
    torch.cuda.set_device(a.device)
    lt = dada.fold_file(a.file, cfg, polyco=polyco, device=a.device, stream=torch.cuda.current_stream().cuda_stream,
                        dump_before=tuple(a.dump))
    for line in lt.vitals():
        print(line, file=sys.stderr)
    for n, sub in enumerate(lt.subints):
        path = "%s_%04d.ps" % (a.prefix, n)
        pipeline.write_phase_series(path, sub, info, cfg, npol=lt.npol_out, scale=lt.scalefac, division=n,
                                    start_seconds=lt.out_start, folding_period=pfold)
        print("dspsr_amd: %s  integration %.6f s  %d samples" % (path, sub["integration_length"], sub["ndat_total"]))
    if a.record:
        lt.report()
    lt.close()


if __name__ == "__main__":
    main()
