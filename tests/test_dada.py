"""dspsr_amd.dada: the DADA header and file layout on the input side of the path (ascii_header.c:31-131,
ASCIIObservation.C:82-415, DADAFile.C:33-180).  Host logic on the CPU; the GPU cases fold a synthetic file end to end and
compare with the same bytes handed over as device-resident blocks."""
import os
import types

import numpy as np
import pytest

from dspsr_amd import dada, synth
from dspsr_amd.engine import DspsrAmdError

REF_HEADER = "/root/reference/Benchmark/header.dada"


def test_header_find_semantics():
    h = "HDR_VERSION 1.0\n# BW 999 in a comment\nXBW 5\nBWX 7\nBW\t-400   # bandwidth in MHz\nFREQ  1382\t\t# centre\nNDIM 1\nBW 64\n"
    assert dada.header_get(h, "HDR_VERSION") == "1.0"            # first word of the header
    assert dada.header_get(h, "BW") == "-400"                     # not the comment, not XBW / BWX, first real match wins
    assert dada.header_get(h, "FREQ") == "1382"
    assert dada.header_get(h, "NPOL") is None
    assert dada.header_get("A 1\\B 2\n", "B") == "2"              # a backslash also counts as a line start
    assert dada.header_get("NBIT 8bits\n", "NBIT") == "8bits" and dada._scan("NBIT 8bits\n", "NBIT", int, 2) == 8
    assert dada.header_get("KEY\n", "KEY") is None                # keyword without the blank that must follow it


def test_observation_defaults_and_errors():
    info, ex = dada.observation("HDR_VERSION 1.0\nTSAMP 0.5\nFREQ 1400\nBW 16\n")
    assert (info.nchan, info.npol, info.ndim, ex["nbit"], info.machine) == (1, 1, 1, 2, "DADA")   # ASCIIObservation defaults
    assert info.rate == 2e6 and ex["ndat"] == 0 and ex["dual_sideband"] is None and info.start_seconds == 0.0
    with pytest.raises(DspsrAmdError, match="invalid NDIM=3"):
        dada.observation("TSAMP 1\nNDIM 3\n")
    with pytest.raises(DspsrAmdError, match="strptime"):
        dada.observation("TSAMP 1\nUTC_START yesterday\n")
    with pytest.raises(DspsrAmdError, match="TSAMP"):
        dada.observation("NDIM 1\n")


def test_observation_matches_the_oracle(oracle):
    raw = synth.dada_header(1382.0, -400.0, 8, 2, 2, 0.02, extra={"OBS_OFFSET": 64000, "DSB": 1, "DM": 67.99})
    # OBS_OFFSET appears twice now (0 first): the first match wins, as in ascii_header_find
    text = raw.split(b"\0", 1)[0].decode()
    info, ex = dada.observation(text)
    assert ex["offset_bytes"] == 0
    text = text.replace("OBS_OFFSET 0\n", "")
    info, ex = dada.observation(text)
    obs = oracle.observation_from_header(oracle.parse_dada_header(text.encode()))
    assert (info.centre_frequency, info.bandwidth, info.nchan, info.npol, info.ndim, ex["nbit"], info.machine) == \
        (obs.centre_frequency, obs.bandwidth, obs.nchan, obs.npol, obs.ndim, obs.nbit, obs.machine)
    assert info.rate == obs.rate and info.start_seconds == obs.start_seconds == 2000 * 0.02e-6
    assert ex["dual_sideband"] is True and ex["dm"] == 67.99
    assert (info.mjd_day, info.mjd_sec) == (55299, 7545.0)        # 2010-04-13-02:05:45


def test_reference_benchmark_header():
    if not os.path.exists(REF_HEADER):
        pytest.skip("reference tree not present (GPU box)")
    text = open(REF_HEADER, "rb").read().split(b"\0", 1)[0].decode("latin-1")
    info, ex = dada.observation(text)
    assert (info.centre_frequency, info.bandwidth, info.nchan, info.npol, info.ndim, ex["nbit"]) == (1382.0, -400.0, 1, 2, 1, 8)
    assert info.machine == "CASPSR" and info.tsamp_us == 0.00125 and ex["source"] == "J0437-4715" and ex["telescope"] == "PKS"
    assert (info.mjd_day, info.mjd_sec) == (55299, 7545.0) and ex["ndat"] == 1024000000000 and ex["resolution"] == 4


def _fake_lt(step, overlap, ppb):
    return types.SimpleNamespace(nsamp_step=step, nsamp_overlap=overlap, cfg=types.SimpleNamespace(parts_per_block=ppb))


def test_file_layout_and_blocks(tmp_path):
    # 2 input channels, complex, dual-pol: 8 bytes per time sample; HDR_SIZE larger than the first read
    hdr = synth.dada_header(1400.0, 32.0, 2, 2, 2, 0.0625, size=8192)
    ndat = 1000
    data = (np.arange(ndat * 8) % 251 - 125).astype(np.int8)
    p = tmp_path / "x.dada"
    p.write_bytes(hdr + data.tobytes() + b"\x01\x02\x03")         # three stray bytes: not a whole sample
    assert dada.is_valid(str(p))
    f = dada.DadaFile(str(p))
    assert (f.header_bytes, f.bytes_per_sample, f.ndat) == (8192, 8, ndat)
    lt = _fake_lt(step=100, overlap=30, ppb=4)
    assert f.nblocks(lt) == (2, 1)                                # (1000-30)//100 = 9 parts = 2 blocks of 4 + 1
    got = list(f.blocks(lt))
    assert [n for _, n in got] == [4, 4, 1]
    for b, (blk, npart) in enumerate(got):
        s0 = b * 400
        assert np.array_equal(blk, data[s0 * 8:(s0 + npart * 100 + 30) * 8])          # consecutive blocks share the overlap
    one = list(f.blocks(lt, channel=1))
    assert np.array_equal(one[2][0], data.reshape(ndat, 2, 4)[800:930, 1, :].reshape(-1))
    assert dada.DadaFile(str(p)).nblocks(_fake_lt(2000, 30, 4)) == (0, 0)
    # header in a separate .hdr file (no HDR_SIZE): the data start at byte 0
    q = tmp_path / "y.dada"
    q.write_bytes(data.tobytes() + b"\0" * 4096)
    text = hdr.split(b"\0", 1)[0].decode().replace("HDR_SIZE 8192\n", "")
    (tmp_path / "y.hdr").write_text(text + "\n")
    g = dada.DadaFile(str(q))
    assert g.header_bytes == 0 and g.info.nchan == 2 and np.array_equal(np.asarray(g._map[:64]), data[:64])
    r = tmp_path / "z.dada"
    r.write_bytes(b"\0" * 100)
    assert not dada.is_valid(str(r))
    with pytest.raises(DspsrAmdError, match="fread"):
        dada.read_header(str(r))
    two = tmp_path / "two.dada"
    two.write_bytes(synth.dada_header(1400.0, 32.0, 1, 2, 1, 0.0625).replace(b"NBIT 8", b"NBIT 2") + b"\0" * 64)
    with pytest.raises(DspsrAmdError, match="NBIT=2"):
        dada.DadaFile(str(two))


@pytest.mark.gpu
def test_fold_file_equals_resident_blocks(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from dspsr_amd import pipeline
    freq, bw, tsamp, dm, period = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004
    cfg = pipeline.Config(nchan=16, dispersion_measure=dm, nbin=64, folding_period=period, ndim=4, parts_per_block=3,
                          max_parts=2)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA", mjd_day=55299, mjd_sec=7545.0)
    stream = torch.cuda.current_stream().cuda_stream
    ref = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
    step, ovl = ref.nsamp_step, ref.nsamp_overlap
    nparts = 3 * 3 + 2                                             # three whole blocks and a ragged one of two parts
    ndat = nparts * step + ovl + 17                                # 17 samples that no part uses
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period)
    path = tmp_path / "synthetic.dada"
    path.write_bytes(synth.dada_header(freq, bw, 1, 2, 1, tsamp) + raw.tobytes())
    dev = torch.from_numpy(raw.reshape(-1)).cuda()
    for b in range(4):
        npart = 3 if b < 3 else 2
        s0 = b * 3 * step
        ref.process_block(dev[2 * s0: 2 * (s0 + npart * step + ovl)], npart)
    ref.finish_subint()
    ref.synchronize()
    want = ref.subints[0]
    lt = dada.fold_file(str(path), cfg, device=0, stream=stream)
    assert (lt.info.centre_frequency, lt.info.bandwidth, lt.info.machine, lt.info.mjd_day) == (freq, bw, "DADA", 55299)
    got = lt.subints[0]
    assert got["ndat_total"] == want["ndat_total"] == nparts * ref.nkeep
    assert np.array_equal(got["hits"], want["hits"]) and torch.equal(got["profile_dev"], want["profile_dev"])
    assert float(got["profile_dev"].abs().max()) > 0
    lt.close()
    ref.close()
    short = tmp_path / "short.dada"
    short.write_bytes(synth.dada_header(freq, bw, 1, 2, 1, tsamp) + raw.tobytes()[:2 * step])
    with pytest.raises(DspsrAmdError, match="fewer than one overlap-save part"):
        dada.fold_file(str(short), cfg, device=0, stream=stream)


def test_dump_header_round_trip():
    h = dada.unload_header(centre_frequency=1382.0, bandwidth=-400.0, nchan=16, npol=2, ndim=2, nbit=32, state="Analytic",
                           rate=25e6, mjd_day=55299, mjd_sec=7545.25, source="J0835-4510", telescope="PKS")
    assert len(h) == 4096
    text = h.split(b"\0", 1)[0].decode()
    assert text.startswith("HDR_VERSION  1.000000               \nTELESCOPE    PKS ")       # "%-12s %-20s   " per key
    info, ex = dada.observation(text)
    assert (info.centre_frequency, info.bandwidth, info.nchan, info.npol, info.ndim, ex["nbit"]) == (1382.0, -400.0, 16, 2, 2, 32)
    assert info.machine == "dspsr" and info.tsamp_us == 0.04 and (info.mjd_day, info.mjd_sec) == (55299, 7545.0)
    assert info.start_seconds == 0.25 and info.source == "J0835-4510"                      # the fraction comes back as OBS_OFFSET
    assert dada.header_set("A 1 # c\nB 2\n", "A", "77") == "A            77                     # c\nB 2\n"
    assert dada.header_set("A 1\nDATA\n", "C", 3).endswith("C            3                      \nDATA\n")


@pytest.mark.gpu
def test_stage_capture_taps(oracle, tmp_path):
    """dspsr --dump Detection --dump Fold: pre_Detection.dump holds the filterbank's complex output, pre_Fold.dump the
    detected samples, both as dsp::Dump writes them (4096-byte header + TFP floats); the tapped run folds the same profile."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from dspsr_amd import pipeline
    freq, bw, tsamp, dm, period = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004
    cfg = pipeline.Config(nchan=16, dispersion_measure=dm, nbin=64, folding_period=period, ndim=4, parts_per_block=3, max_parts=2,
                          force_fused=True)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA", source="J0835-4510")
    stream = torch.cuda.current_stream().cuda_stream
    plain = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
    tapped = pipeline.LoadToFold(cfg, info, device=0, stream=stream, dump_before=("Detection", "Fold"), dump_dir=str(tmp_path))
    assert plain.fused_fold and not tapped.fused_fold
    step, ovl, nkeep = plain.nsamp_step * 3, plain.nsamp_overlap, plain.nkeep
    raw = synth.voltages(2 * step + ovl, freq, bw, tsamp, dm, period)
    dev = torch.from_numpy(raw.reshape(-1)).cuda()
    for lt in (plain, tapped):
        for b in range(2):
            lt.process_block(dev[2 * b * step: 2 * (b * step + step + ovl)])
        lt.finish_subint()
        lt.synchronize()
    assert np.array_equal(plain.subints[0]["hits"], tapped.subints[0]["hits"])
    assert torch.equal(plain.subints[0]["profile_dev"], tapped.subints[0]["profile_dev"])
    detected_last = tapped.detected.clone()
    out_rate, out_start = tapped.out_rate, tapped.out_start
    tapped.close()
    plain.close()
    text, cplx = dada.read_dump(str(tmp_path / "pre_Detection.dump"))
    hinfo, ex = dada.observation(text)
    assert (hinfo.nchan, hinfo.npol, hinfo.ndim, ex["nbit"], hinfo.machine) == (16, 2, 2, 32, "dspsr")
    assert dada.header_get(text, "STATE") == "Analytic" and hinfo.source == "J0835-4510"
    assert abs(hinfo.rate - out_rate) <= 1e-6 * out_rate and abs(hinfo.start_seconds - out_start) <= 1.5 / out_rate   # OBS_OFFSET holds whole samples (truncated, ASCIIObservation.C:566-567)
    assert cplx.shape == (2 * 3 * nkeep, 16, 2, 2)
    text, det = dada.read_dump(str(tmp_path / "pre_Fold.dump"))
    hinfo, ex = dada.observation(text)
    assert (hinfo.nchan, hinfo.npol, hinfo.ndim) == (16, 1, 4) and dada.header_get(text, "STATE") == "Coherence"
    assert det.shape == (2 * 3 * nkeep, 16, 1, 4)
    # the second block of the detected dump is what the pipeline held in HBM; detection of the complex dump gives it back
    last = detected_last.view(16, 1, 3 * nkeep, 4).permute(2, 0, 1, 3).cpu().numpy()
    assert np.array_equal(det[3 * nkeep:], last)
    z = cplx[..., 0].astype(np.float64) + 1j * cplx[..., 1]
    prod = oracle.detect_products(np.moveaxis(z, 0, 2), "Coherence")             # [chan][4][ndat]
    want = np.moveaxis(prod, 1, 2)                                                # [chan][ndat][4]
    got = np.moveaxis(det[:, :, 0, :], 0, 1)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    with pytest.raises(DspsrAmdError, match="no operation named"):
        pipeline.LoadToFold(cfg, info, device=0, stream=stream, dump_before=("Filterbank",))


@pytest.mark.gpu
def test_record_time_report(capsys):
    """dspsr -r: every operation timed with a stream synchronisation behind it (Operation.C:90-113), the table of
    Operation::report; the results do not change."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import io
    from dspsr_amd import pipeline
    freq, bw, tsamp, dm, period = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    stream = torch.cuda.current_stream().cuda_stream
    profs = []
    for rec, fused in ((False, False), (True, False), (True, True)):
        cfg = pipeline.Config(nchan=16, dispersion_measure=dm, nbin=64, folding_period=period, ndim=4, parts_per_block=3,
                              max_parts=2, record_time=rec, fused_fold=fused, force_fused=fused)
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
        raw = torch.from_numpy(synth.voltages(lt.nsamp_step * 3 + lt.nsamp_overlap, freq, bw, tsamp, dm, period)).cuda()
        lt.process_block(raw)
        lt.finish_subint()
        profs.append(lt.subints[0]["profile_dev"].clone())
        assert lt.vitals()[0].startswith("dspsr: dedispersion filter length=%d (minimum=" % lt.response.ndat)
        out = io.StringIO()
        lt.report(out)
        text = out.getvalue()
        if not rec:
            assert text == "" and lt.optime == {}
        elif fused:
            assert list(lt.optime) == ["Filterbank+Detection+Fold"] and lt.optime["Filterbank+Detection+Fold"][1] == 1
        else:
            assert list(lt.optime) == ["Filterbank+Detection", "Fold"] and all(t > 0 for t, _ in lt.optime.values())
            assert text.splitlines()[0].split() == ["Operation", "Time", "Spent", "Discarded"] and "Fold" in text
        lt.close()
    assert torch.equal(profs[0], profs[1]) and torch.equal(profs[0], profs[2])


@pytest.mark.gpu
def test_dspsr_spelled_driver(tmp_path):
    """tools/dspsr_amd_fold.py with the reference's option spellings on a synthetic DADA file: sub-integrations of 2 ms land
    in PhaseSeries hand-off files that read back with the right shape and sample counts."""
    import subprocess
    import sys
    from dspsr_amd import pipeline
    freq, bw, tsamp, dm, period = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004
    raw = synth.voltages(400000, freq, bw, tsamp, dm, period)
    path = tmp_path / "synthetic.dada"
    path.write_bytes(synth.dada_header(freq, bw, 1, 2, 1, tsamp) + raw.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "dspsr_amd_fold.py"), "-F", "16:D", "-D", str(dm), "-b", "64", "-c", str(period),
           "-L", "0.002", "-r", "-O", str(tmp_path / "out"), str(path)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "dspsr: dedispersion filter length=" in p.stderr and "Time Spent" in p.stderr
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("out_") and f.endswith(".ps"))
    assert len(files) >= 5
    total = 0
    for f in files:
        hdr, hits, prof = pipeline.read_phase_series(str(tmp_path / f))
        assert prof.shape == (16, 1, 64, 4) and int(hits.sum()) == int(hdr["NDAT_TOTAL"])
        total += int(hdr["NDAT_TOTAL"])
    assert total > 0 and float(np.abs(prof).max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("fb,dm", [("16", 30.0), ("16", 0.0)])
def test_dspsr_spelled_driver_filterbank_then_convolution(tmp_path, fb, dm):
    """`-F 16` without `:D` (Filterbank::Config::After; -D 0: the filterbank alone) through the same tool: the non-convolving filterbank
    and dsp::Convolution on its channels, fed from the file block by block."""
    import subprocess
    import sys
    from dspsr_amd import pipeline
    freq, bw, tsamp, period = 1382.0, -16.0, 1.0 / 32.0, 0.004
    raw = synth.voltages(400000, freq, bw, tsamp, max(dm, 1.0), period)
    path = tmp_path / "synthetic.dada"
    path.write_bytes(synth.dada_header(freq, bw, 1, 2, 1, tsamp) + raw.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "dspsr_amd_fold.py"), "-F", fb, "-D", str(dm), "-b", "64", "-c", str(period),
           "-L", "0.002", "-O", str(tmp_path / "out"), str(path)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("out_") and f.endswith(".ps"))
    assert len(files) >= 5
    total = 0
    for f in files:
        hdr, hits, prof = pipeline.read_phase_series(str(tmp_path / f))
        assert prof.shape == (16, 1, 64, 4) and int(hits.sum()) == int(hdr["NDAT_TOTAL"])
        total += int(hdr["NDAT_TOTAL"])
    # one output sample per 32 input samples, less the convolution's overlap at the end of the file
    assert 0 < total <= 400000 // 32 and total >= 400000 // 32 - 2200 and float(np.abs(prof).max()) > 0
