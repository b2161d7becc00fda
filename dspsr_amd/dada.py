"""DADA files on the input side of the path: the ASCII header (Kernel/Classes/ascii_header.c:31-131,
ASCIIObservation.C:82-415) and the file layout (DADAFile.C:33-180), read the way the reference's dsp::DADAFile +
dsp::IOManager hand blocks to the pipeline -- consecutive blocks of parts_per_block overlap-save parts that share
nsamp_overlap samples (the InputBuffering tail, Filterbank.C:443-444).

Host logic only (no arithmetic of the path happens here): the blocks it yields are pinned host tensors for
pipeline.LoadToFold.process_host_blocks, which copies them to the device while the previous block is processed.
"""
from __future__ import annotations

import calendar
import math
import os
import time

import numpy as np

from .engine import DspsrAmdError

DEFAULT_HEADER_SIZE = 4096            # DADAFile.C:44


def header_find(header: str, keyword: str) -> int:
    """ascii_header_find (ascii_header.c:31-51): the first occurrence of `keyword` that is the first word of the header or
    is preceded by a newline (or a backslash) AND is followed by a blank or a tab.  Returns the index or -1."""
    n = len(keyword)
    k = header.find(keyword)
    while k > 0:
        nxt = header[k + n] if k + n < len(header) else ""
        if header[k - 1] in "\n\\" and nxt in ("\t", " "):
            break
        k = header.find(keyword, k + 1)
    return k


def header_get(header: str, keyword: str):
    """ascii_header_get with a one-token format (%s, %d, %lf ...): the first white-space delimited token after the keyword,
    or None when the keyword is absent.  A trailing '# comment' is never part of the value."""
    k = header_find(header, keyword)
    if k < 0:
        return None
    rest = header[k:]
    i = 0
    while i < len(rest) and rest[i] not in " \t\n":          # value = key + strcspn(key, whitespace)
        i += 1
    tok = rest[i:].split(None, 1)
    return tok[0] if tok else None


def _scan(header, keyword, conv, default):
    v = header_get(header, keyword)
    if v is None:
        return default
    try:
        return conv(v)
    except ValueError:
        # sscanf reads the longest valid prefix ("8bits" -> 8); mirror it for the numeric conversions
        import re
        m = re.match(r"[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", v)
        if not m:
            return default
        return conv(m.group(0))


def read_header(path: str):
    """DADAFile::get_header (DADAFile.C:33-110): read 4096 bytes, grow to HDR_SIZE if the header says it is larger; a file
    without HDR_SIZE takes its header from a matching .hdr file (then the data start at byte 0).
    Returns (header text, header_bytes)."""
    size = DEFAULT_HEADER_SIZE
    with open(path, "rb") as f:
        while True:
            f.seek(0)
            buf = f.read(size)
            if len(buf) != size:
                raise DspsrAmdError("dsp::DADAFile::get_header fread (nbyte=%d)" % size)
            text = buf[:-1].split(b"\0", 1)[0].decode("latin-1")      # header[hdr_size-1] = '\0'
            hs = _scan(text, "HDR_SIZE", int, 0)
            if hs <= size:
                break
            size = hs
    if hs:
        return text, hs
    base = os.path.splitext(path)[0] + ".hdr"
    if not os.path.exists(base):
        base = path + ".hdr"
    if not os.path.exists(base):
        raise DspsrAmdError("dsp::DADAFile::get_header file has no header and no matching header file found")
    with open(base, "rb") as f:
        buf = f.read()
    return buf[:-1].split(b"\0", 1)[0].decode("latin-1"), 0


def is_valid(path: str) -> bool:
    """DADAFile::is_valid (DADAFile.C:112-147): HDR_VERSION and INSTRUMENT must be present."""
    try:
        text, _ = read_header(path)
    except (OSError, DspsrAmdError):
        return False
    return bool(text) and header_get(text, "HDR_VERSION") is not None and header_get(text, "INSTRUMENT") is not None


def observation(header: str):
    """ASCIIObservation::load (ASCIIObservation.C:82-415), the keys this path uses, with the reference's defaults
    (NCHAN 1, NPOL 1, NBIT 2, NDIM 1) and its errors.  Returns (pipeline.InputInfo, extras) where extras holds
    nbit, ndat (0 = take it from the file size), offset_bytes, source, telescope, dual_sideband (None = by state) and dm."""
    from .pipeline import InputInfo
    if header_get(header, "HDR_VERSION") is None and header_get(header, "CPSR2_HEADER_VERSION") is None:
        pass                                                           # the reference only warns (:97-104)
    nchan = _scan(header, "NCHAN", int, 1)
    npol = _scan(header, "NPOL", int, 1)
    nbit = _scan(header, "NBIT", int, 2)
    ndim = _scan(header, "NDIM", int, 1)
    if ndim not in (1, 2, 4):
        raise DspsrAmdError("ASCIIObservation invalid NDIM=%d" % ndim)
    tsamp = _scan(header, "TSAMP", float, 0.0)
    if tsamp <= 0:
        raise DspsrAmdError("ASCIIObservation TSAMP missing or not positive")       # rate = 1/0 in the reference
    day, sec = 0, 0.0
    utc = header_get(header, "UTC_START")
    if utc is not None:
        try:
            t = calendar.timegm(time.strptime(utc, "%Y-%m-%d-%H:%M:%S"))
        except ValueError:
            raise DspsrAmdError("ASCIIObservation failed strptime (%s)" % utc)
        day, sec = 40587 + t // 86400, float(t % 86400)                # MJD(time_t): 1970-01-01 = MJD 40587
        sec += _scan(header, "PICOSECONDS", int, 0) / 1e12
    off = header_get(header, "OBS_OFFSET")
    if off is None:
        off = header_get(header, "OFFSET")
    offset_bytes = int(off) if off is not None else 0
    bits_per_samp = nchan * npol * ndim * nbit
    start = 0.0
    if utc is not None:
        start = float((offset_bytes * 8) // bits_per_samp) * (tsamp * 1e-6)          # get_nsamples(offset_bytes) * tsamp
    machine = header_get(header, "INSTRUMENT") or "DADA"
    dsb = header_get(header, "DSB")
    info = InputInfo(centre_frequency=_scan(header, "FREQ", float, 0.0), bandwidth=_scan(header, "BW", float, 0.0),
                     nchan=nchan, npol=npol, ndim=ndim, tsamp_us=tsamp, machine=machine, start_seconds=start,
                     mjd_day=day, mjd_sec=sec, source=header_get(header, "SOURCE") or "unknown",
                     telescope=header_get(header, "TELESCOPE") or "unknown")
    extras = {"nbit": nbit, "ndat": _scan(header, "NDAT", int, 0), "offset_bytes": offset_bytes,
              "source": header_get(header, "SOURCE") or "unknown", "telescope": header_get(header, "TELESCOPE") or "unknown",
              "dual_sideband": None if dsb is None else int(dsb) == 1, "dm": _scan(header, "DM", float, None),
              "resolution": _scan(header, "RESOLUTION", int, 1)}
    return info, extras


def header_set(header: str, keyword: str, value) -> str:
    """ascii_header_set (ascii_header.c:53-108): replace the value of an existing keyword (up to its comment or line end)
    or append 'KEYWORD      value                  ' + newline (in front of a "DATA" line if there is one)."""
    text = "%-12s %-20s   " % (keyword, value)
    k = header_find(header, keyword)
    if k >= 0:
        e = k
        while e < len(header) and header[e] not in "#\n":
            e += 1
        return header[:k] + text + header[e:]
    d = header.find("DATA\n")
    if d >= 0:
        return header[:d] + text + "\n" + header[d:]
    return header + text + "\n"


def unload_header(*, centre_frequency, bandwidth, nchan, npol, ndim, nbit, state, rate, mjd_day, mjd_sec, machine="dspsr",
                  telescope="unknown", receiver="unknown", source="unknown", size=DEFAULT_HEADER_SIZE) -> bytes:
    """ASCIIObservation::unload (ASCIIObservation.C:423-582) + HDR_SIZE, as dsp::Dump::prepare writes it in front of a
    binary dump (Dump.C:46-66): the keys in the reference's order, values in its printf formats, zero padded to `size`."""
    whole = int(math.floor(mjd_sec))
    day, sec = mjd_day + whole // 86400, whole % 86400
    t = time.gmtime((day - 40587) * 86400 + sec)
    frac = mjd_sec - whole
    bits = nchan * npol * ndim * nbit
    offset_bytes = (int(frac * rate) * bits) // 8                       # get_nbytes(offset_samples)
    h = ""
    for key, val in (("HDR_VERSION", "%f" % 1.0), ("TELESCOPE", telescope), ("RECEIVER", receiver), ("SOURCE", source),
                     ("MODE", "PSR"), ("FREQ", "%f" % centre_frequency), ("BW", "%f" % bandwidth), ("NCHAN", "%d" % nchan),
                     ("NPOL", "%d" % npol), ("NBIT", "%d" % nbit), ("NDIM", "%d" % ndim), ("STATE", state),
                     ("TSAMP", "%f" % (1e6 / rate)), ("UTC_START", time.strftime("%Y-%m-%d-%H:%M:%S", t)),
                     ("OBS_OFFSET", "%d" % offset_bytes), ("INSTRUMENT", machine), ("HDR_SIZE", "%d" % size)):
        h = header_set(h, key, val)
    raw = h.encode("ascii")
    if len(raw) >= size:
        raise DspsrAmdError("dspsr_amd.dada.unload_header: header of %d bytes does not fit HDR_SIZE %d" % (len(raw), size))
    return raw + b"\0" * (size - len(raw))


class Dump:
    """dsp::Dump in binary mode (Dump.C:46-99; inserted by `dspsr --dump <Operation>`, SingleThread.C:315-346, as
    pre_<Operation>.dump): a 4096-byte DADA header of the TimeSeries, then for every call the block's samples in TFP order
    (for idat, ichan, ipol: ndim floats).  The reference's only stage-capture mechanism; the taps LoadToFold offers are the
    same two, "Detection" (the filterbank's complex output) and "Fold" (the detected samples)."""

    def __init__(self, path, **header_fields):
        self.path = path
        self.f = open(path, "wb")
        self.f.write(unload_header(**header_fields))
        self.ndat = 0

    def write(self, rows, ndat, ndim):
        """rows: torch tensor [nchan][npol][>= ndat*ndim] float32 (device or host), FPT order as in the TimeSeries."""
        nchan, npol = rows.shape[0], rows.shape[1]
        tfp = rows[:, :, :ndat * ndim].reshape(nchan, npol, ndat, ndim).permute(2, 0, 1, 3).contiguous()
        tfp.cpu().numpy().tofile(self.f)
        self.ndat += ndat

    def close(self):
        if self.f:
            self.f.close()
            self.f = None


def read_dump(path):
    """(header text, float32 array [ndat][nchan][npol][ndim]) of a binary dsp::Dump file."""
    text, hb = read_header(path)
    info, ex = observation(text)
    data = np.fromfile(path, dtype=np.float32, offset=hb)
    per = info.nchan * info.npol * info.ndim
    return text, data[:(data.size // per) * per].reshape(-1, info.nchan, info.npol, info.ndim)


class DadaFile:
    """dsp::DADAFile (DADAFile.C:149-180) + the block loop of dsp::IOManager: an 8-bit DADA file cut into the blocks
    LoadToFold consumes."""

    def __init__(self, path: str):
        self.path = path
        self.header, self.header_bytes = read_header(path)
        self.info, self.extras = observation(self.header)
        if self.extras["nbit"] != 8:
            raise DspsrAmdError("dspsr_amd.DadaFile: NBIT=%d; this path reads 8-bit samples" % self.extras["nbit"])
        self.bytes_per_sample = self.info.nchan * self.info.npol * self.info.ndim
        nbytes = os.path.getsize(path) - self.header_bytes
        self.ndat = nbytes // self.bytes_per_sample                     # File::open_fd: from the file size
        self._map = np.memmap(path, dtype=np.int8, mode="r", offset=self.header_bytes,
                              shape=(self.ndat * self.bytes_per_sample,)) if self.ndat else np.zeros(0, np.int8)

    def nblocks(self, lt):
        """(whole blocks, parts of the ragged last block)."""
        if self.ndat < lt.nsamp_overlap + lt.nsamp_step:
            return 0, 0
        nparts = (self.ndat - lt.nsamp_overlap) // lt.nsamp_step
        return divmod(nparts, lt.cfg.parts_per_block)

    def blocks(self, lt, channel=None):
        """Yield (int8 host array, npart) for LoadToFold `lt`; the last block may hold fewer parts.  Samples past the last
        whole overlap-save part are not used (the reference leaves them in the input buffer at end of data).  The arrays
        are views of the memory-mapped file (process_host_blocks stages them through its own pinned buffers).
        channel = g: only input channel g's bytes ([t][pol][dim]), what rank g of a sub-band sharded run reads."""
        full, rest = self.nblocks(lt)
        ppb, step, ovl, bps = lt.cfg.parts_per_block, lt.nsamp_step, lt.nsamp_overlap, self.bytes_per_sample
        per_chan = self.info.npol * self.info.ndim
        for b in range(full + (1 if rest else 0)):
            npart = ppb if b < full else rest
            nsamp = npart * step + ovl
            s0 = b * ppb * step
            src = self._map[s0 * bps:(s0 + nsamp) * bps]
            if channel is not None:
                src = np.ascontiguousarray(src.reshape(nsamp, self.info.nchan, per_chan)[:, channel, :]).reshape(-1)
            yield src, npart


def fold_file(path, cfg, polyco=None, device=0, stream=None, reference_phase=0.0, dump_before=(), dump_dir="."):
    """The reference's `dspsr file.dada -F nchan:D ...` on one GPU: open the file, build the pipeline from its header,
    feed every block, close the last sub-integration.  Returns the LoadToFold (its .subints hold the results; the caller
    closes it)."""
    from .pipeline import LoadToFold
    f = DadaFile(path)
    lt = LoadToFold(cfg, f.info, device=device, stream=stream, polyco=polyco, reference_phase=reference_phase,
                    dump_before=dump_before, dump_dir=dump_dir)
    if f.nblocks(lt) == (0, 0):
        lt.close()
        raise DspsrAmdError("dspsr_amd.fold_file: %s holds %d samples, fewer than one overlap-save part (%d)"
                            % (path, f.ndat, lt.nsamp_overlap + lt.nsamp_step))
    lt.process_host_blocks(f.blocks(lt))
    if lt.ndat_total:
        lt.finish_subint()
    lt.synchronize()
    return lt
