"""K identical blocks of one bench.py workload and nothing else -- the program rocprofv3 --pmc wraps (tools/pmc_traffic.sh).

    python tools/pmc_workload.py WORKLOAD fused|unfused [K]

The pipeline object, the synthetic block and the launch shape are bench.py's own (bench.make_fold_pipeline /
bench_search_mode / bench_fold_only set-ups), so the counters belong to exactly the launches the bench times.  No warm-up
run, no parity gate, no oracle: every block launches the same kernels, so a counter summed over the run and divided by K
is the traffic of one block.  Prints one JSON line with the geometry the post-processing needs.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, mode = sys.argv[1], sys.argv[2]
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    import numpy as np
    import torch
    import bench
    wl = bench.WORKLOADS[name]
    args = argparse.Namespace(parts_per_block=0, max_parts=0, ndim=4, no_fused_fold=(mode != "fused"), dump_steps=8)
    import dspsr_amd
    rec = {"workload": name, "mode": mode, "blocks": K, "build_id": dspsr_amd.build_id()}
    if name == "cfg5":
        from dspsr_amd import pipeline
        info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=1, npol=2, ndim=1, tsamp_us=wl["tsamp_us"],
                                  machine=wl["machine"])
        lf = pipeline.LoadToFil(pipeline.SearchConfig(nchan=wl["nchan"], tscrunch=wl["tscrunch"], nbit=8, parts_per_block=wl["nparts"]),
                                info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        nbytes = lf.block_bytes()
        raw = torch.randn(nbytes, generator=torch.Generator(device="cuda").manual_seed(20100413), device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
        torch.cuda.synchronize()
        for _ in range(K):
            lf.process_block(raw)
        torch.cuda.synchronize()
        rec.update(parts_per_block=wl["nparts"], parts_per_launch_group=wl["nparts"],
                   algorithmic_bytes_per_block=nbytes + lf.detected.numel() * 4, roofline_kernel="k_tfp")
        lf.close()
    elif name == "cfg5c":
        lf, raw, cfg = bench.make_search_coherent(wl, torch)
        torch.cuda.synchronize()
        for _ in range(K):
            lf.detect_scrunch(raw)                 # the roofline group alone (Rescale + digitiser run on the 16x smaller rows)
        torch.cuda.synchronize()
        N = cfg.nchan * lf.response.ndat
        b_alg = 2 * (2 * N) + 8 * N + cfg.nchan * cfg.npol * 4 * lf.nkeep / cfg.tscrunch
        rec.update(parts_per_block=cfg.parts_per_block, parts_per_launch_group=cfg.max_parts, algorithmic_bytes_per_part=int(b_alg),
                   algorithmic_bytes_per_block=int(b_alg * cfg.parts_per_block))
        lf.close()
    elif name == "fold":
        import dspsr_amd
        from dspsr_amd import pipeline
        nchan, npol, ndat = wl["nchan"], wl["npol"], wl["ndat"]
        rate = 1e6 / wl["tsamp_us"]
        text = json.load(open(os.path.join(ROOT, "tests", "golden", "vela_polyco.json")))["text"]
        polyco = pipeline.Polyco(text)
        day, sec0 = 55299, 7545.0
        nbin = pipeline.choose_nbin(1.0 / polyco.frequency(day, sec0), rate)
        ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
        fold = dspsr_amd.FoldEngine(ctx)
        fold.set_shape(nchan, npol, 1, nbin)
        det = torch.rand((nchan, npol, ndat), dtype=torch.float32, device="cuda")
        hits = np.zeros(nbin, np.uint32)
        torch.cuda.synchronize()
        for b in range(K):
            t0 = (b * ndat + 0.5) / rate
            fold.set_nbin(nbin)
            fold.set_ndat(ndat, 0)
            fold.set_bins(polyco.phase_frac(day, sec0 + t0), (1.0 / rate) * polyco.frequency(day, sec0 + t0), ndat, 0, hits)
            fold.fold(det)
        torch.cuda.synchronize()
        rec.update(parts_per_block=1, parts_per_launch_group=1, algorithmic_bytes_per_block=det.numel() * 4, roofline_kernel="k_fold_")
        fold.close()
        ctx.close()
    else:
        lt, raw, cfg, info, wl, sharded = bench.make_fold_pipeline(name, args, torch, 0, 1, 0)
        torch.cuda.synchronize()
        for _ in range(K):
            lt.process_block(raw)
        torch.cuda.synchronize()
        r = lt.response
        nchan_subband = cfg.nchan // info.nchan
        N = nchan_subband * (r.ndat if r is not None else 1)
        nsamp_fft = 2 * N if info.ndim == 1 else N
        b_alg = bench.algorithmic_bytes_per_part(2, nsamp_fft * info.ndim, 8, N, nchan_subband, lt.nkeep)
        if r is None:
            b_alg -= 8 * N                          # Config::Never: no response to read
        if mode == "fused":
            b_alg -= 2 * nchan_subband * lt.nkeep * 8
        rec.update(parts_per_block=cfg.parts_per_block, parts_per_launch_group=cfg.max_parts, fused_fold=bool(lt.fused_fold),
                   algorithmic_bytes_per_part=b_alg, algorithmic_bytes_per_block=b_alg * cfg.parts_per_block * lt.in_nchan)
        lt.close()
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
