"""tests/fuzz_search.py [ncases] [seed] [case,case,...] : random geometries of digifil's convolving branch (dspsr_amd_filterbank_perform_search) -- the
square-law + time-scrunch epilogue of the inverse pass must equal the oracle's Detection::square_law + TScrunch (float32, sequential
sums) applied to the SAME object's complex output BIT FOR BIT, as a stream over several calls of random part counts, for Intensity
and PPQQ, any scrunch factor, real / complex input, 1 or 2 polarisations, several input channels, three-pass / two-pass /
four-pass / odd-factor geometries, launch groups that split a call.  Test infrastructure (uses the oracle)."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import importlib.util

spec = importlib.util.spec_from_file_location("oracle_mod", os.path.join(ROOT, "oracle", "dspsr_oracle.py"))
oracle = importlib.util.module_from_spec(spec)
sys.modules["oracle_mod"] = oracle
spec.loader.exec_module(oracle)
import dspsr_amd

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else None       # replay: run these case numbers of the sequence only
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
bad = 0
for i in range(ncases):
    logN = int(rng.integers(8, 20))
    logM = int(rng.integers(3, min(logN - 1, 15) + 1))
    C, M = 1 << (logN - logM), 1 << logM
    real = bool(rng.integers(0, 3) != 0)
    npol = 2 if rng.integers(0, 4) else 1
    input_nchan = int(rng.choice([1, 1, 1, 2, 3])) if logN <= 17 else 1
    kind = int(rng.integers(0, 8))
    force = 0
    if kind == 0 and logN - logM >= 6:                       # nchan_subband with an odd factor
        R = int(rng.choice([3, 5, 7, 11, 15]))
        C = R * max(1, C >> int(np.ceil(np.log2(R))))
    elif kind == 1 and logM >= 6:                            # freq_res with an odd factor
        R = int(rng.choice([3, 5, 9, 13]))
        M = R * max(2, M >> int(np.ceil(np.log2(R))))
    elif kind == 2:                                          # two-pass family (complex dual-pol 8-bit)
        logM = int(rng.integers(9, 12))
        logC = int(rng.integers(13 - logM, min(27 - 2 * logM, 8) + 1))
        C, M, real, npol, input_nchan = 1 << logC, 1 << logM, False, 2, 1
    elif kind == 3:
        force = 1                                            # four-pass kernels: the operations one after the other
    pos = int(rng.integers(0, max(1, M // 3)))
    neg = int(rng.integers(0, max(1, M // 3)))
    nchan = C * input_nchan
    sf = int(rng.choice([1, 2, 3, 7, 16, 16, 33, 100, int(rng.integers(1, 3000))]))
    state_name = "PPQQ" if (npol == 2 and rng.integers(0, 2)) else "Intensity"
    npo = 2 if state_name == "PPQQ" else 1
    state = dspsr_amd.PPQQ if npo == 2 else dspsr_amd.INTENSITY
    maxp = int(rng.integers(1, 6))
    parts = [int(rng.integers(1, 7)) for _ in range(int(rng.integers(1, 4)))]
    desc = "C=%d M=%d nfilt=(%d,%d) real=%d npol=%d in_nchan=%d sf=%d %s max_parts=%d parts=%s force=%d" % (
        C, M, pos, neg, real, npol, input_nchan, sf, state_name, maxp, parts, force)
    kseed = int(rng.integers(1, 1 << 30))
    if only is not None and i not in only:
        continue
    desc = "#%d %s kseed=%d" % (i, desc, kseed)
    try:
        krng = np.random.default_rng(kseed)
        kernel = np.exp(1j * krng.uniform(-np.pi, np.pi, input_nchan * C * M)).astype(np.complex64)
        fb = dspsr_amd.FilterbankEngine(ctx).setup(C, M, pos, neg, input_nchan, npol, real, kernel, max_parts=maxp, force_four_pass=force)
        carry = torch.zeros((nchan, npo), dtype=torch.float32, device="cuda")
        cc, got, dets = 0, [], []
        for npart in parts:
            nsamp = npart * fb.nsamp_step + fb.nsamp_overlap
            raw = torch.from_numpy(np.clip(np.rint(krng.standard_normal(nsamp * input_nchan * npol * (1 if real else 2)) * 24.0), -128, 127)
                                   .astype(np.int8)).cuda()
            cplx = torch.zeros((nchan, npol, 2 * npart * fb.nkeep), dtype=torch.float32, device="cuda")
            fb.perform_raw(raw, dspsr_amd.RAW_GENERIC, 0.0123, cplx, npart)
            c = cplx.cpu().numpy().view(np.complex64)
            if npol == 1:
                dets.append((c.real * c.real + c.imag * c.imag).astype(np.float32))
            else:
                dets.append(oracle.square_law(c, state_name))
            out = torch.full((nchan, npo, (cc + npart * fb.nkeep) // sf + 1), -1.0, dtype=torch.float32, device="cuda")
            nout, cc = fb.perform_search(out, carry, cc, npart, sf, state, raw=raw, layout=dspsr_amd.RAW_GENERIC, scale=0.0123)
            got.append(out[:, :, :nout].cpu().numpy())
        fused = fb.search_is_fused()
        fb_nkeep = fb.nkeep
        fb.close()
        all_det = np.concatenate(dets, axis=2)
        want = oracle.tscrunch_fpt(all_det, sf) if sf > 1 else all_det
        got = np.concatenate(got, axis=2)
        assert got.shape == want.shape, (got.shape, want.shape)
        assert cc == all_det.shape[2] % sf
        if not np.array_equal(got, want):
            bad_el = np.argwhere(got != want)
            rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
            if os.environ.get("FUZZ_SEARCH_DUMP"):
                cat = np.concatenate([d for d in dets], axis=2)
                for (a, b, c_) in bad_el[:8]:
                    print("   mismatch chan %d pol %d out %d: got %.9g want %.9g (sf %d, first stream sample %d, nkeep %d)" % (
                        a, b, c_, got[a, b, c_], want[a, b, c_], sf, c_ * sf, fb_nkeep), flush=True)
            raise AssertionError("max diff %g at %s; %d of %d elements differ, max relative %.3g; channels %s; samples %d..%d" % (
                np.abs(got - want).max(), np.unravel_index(np.argmax(np.abs(got - want)), got.shape), len(bad_el), got.size, rel.max(),
                sorted(set(bad_el[:, 0].tolist()))[:12], bad_el[:, 2].min(), bad_el[:, 2].max()))
        print("ok   ", desc, "fused=%d" % fused, flush=True)
    except dspsr_amd.DspsrAmdError as e:
        print("refused", desc, "--", str(e)[:100], flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL ", desc, "--", str(e)[:400], flush=True)
    except Exception:
        bad += 1
        print("ERROR", desc, flush=True)
        traceback.print_exc()
ctx.close()
print("%d cases, %d failures" % (ncases, bad))
sys.exit(1 if bad else 0)
