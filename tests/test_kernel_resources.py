"""Register spills are a silent regression: in round 5 an unrelated rewrite of the wgfft driver tipped the fused inverse pass
k_inv_chan<12, 1, 2> -- the headline's largest kernel -- from 0 to 28 bytes of scratch per lane and cost the headline 5 %
(profiles/r05_experiments.txt item 7).  This test reads the AMDGPU metadata of the code objects inside the SHIPPED library (the
clang offload bundles of every translation unit, NT_AMDGPU_METADATA msgpack notes) and requires the kernels the BASELINE workloads
launch to use no scratch memory.  CPU only: nothing is launched."""
import os
import struct
import subprocess

import msgpack
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dspsr_amd", "libdspsr_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    i = blob.find(MAGIC)
    while i >= 0:
        n = struct.unpack_from("<Q", blob, i + len(MAGIC))[0]
        off = i + len(MAGIC) + 8
        for _ in range(n):
            o, s, l = struct.unpack_from("<QQQ", blob, off)
            off += 24
            name = blob[off:off + l]
            off += l
            if name.startswith(b"hip") and s:
                yield blob[i + o:i + o + s]
        i = blob.find(MAGIC, i + 1)


def _kernels(elf):
    """{mangled name: metadata map} from the NT_AMDGPU_METADATA note (type 32, owner AMDGPU) of one ELF64 code object."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    out = {}
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        o, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = o, o + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for kd in md.get("amdhsa.kernels", []):
                    out[kd[".name"]] = kd
    return out


@pytest.fixture(scope="module")
def kernels():
    blob = open(LIB, "rb").read()
    ks = {}
    for co in _code_objects(blob):
        ks.update(_kernels(co))
    assert len(ks) > 100, "no kernel metadata found in %s" % LIB
    names = sorted(ks)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {d.split("(")[0].replace("void dspsr_amd::", "").replace("dspsr_amd::", ""): ks[n] for d, n in zip(dem, names)}


# the kernels the bench workloads launch (profiles/r05z_*_kernel_stats.txt), by workload
HOT = {
    "target / cfg2 / cfg3 / cfg5c": ["k_fwd_cols<12, 1, 2>", "k_fwd_rows<11, 3>", "k_fwd_rows<9, 5>", "k_inv_chan<12, 1, 2>", "k_inv_chan<12, 0, 2>",
                                     "k_inv_chan<12, 2, 2>", "k_raw_transpose"],
    "cfg4": ["k_raw_cols", "k_fwd_col1q<1>", "k_rows_inv<9, 4, 1>", "k_rows_inv<9, 4, 0>", "k_rows_inv<9, 4, 2>"],
    "cfg1 / cfg1opt": ["k_fwd_cols<11, 1, 3>", "k_fwd_rows<10, 4>", "k_fwd_cols_dual<1>", "k_fwd_rows<12, 2>", "k_inv_a<6, false, true, true>",
                       "k_inv_a<10, true, true, true>", "k_inv_b<8, false, true>", "k_inv_b<8, true, true>"],
    "cfg5": ["k_tfp4k<false, false>", "k_tfp4k<false, true>", "k_tfp4k<true, false>", "k_tfp4k<true, true>",
             "k_tfpm<10, true, true>", "k_tfpm<11, true, true>", "k_tfpm<13, true, true>", "k_tfpm<9, true, true>"],
    "fold": ["k_fold_dense<1, 4>", "k_fold_dense<4, 1>"],
    "after / after8k / plain": ["k_fb_plain<7, 0>", "k_fb_plain<10, 0>", "k_fb_plain<7, 2>", "k_conv1<13>", "k_conv1<12>", "k_conv1<10>", "k_conv3_a<16>", "k_conv3_b<16>", "k_conv3_c<16, 1>", "k_conv3_c<16, 0>",
                                "k_conv3_a<14>", "k_conv3_b<14>", "k_conv3_b<17>", "k_conv3_c<17, 1>", "k_conv3_a<19>", "k_conv3_b<19>", "k_conv3_c<19, 1>", "k_conv3_b<20>",
                                "k_conv3_a<21>", "k_conv3_b<21>", "k_conv3_c<21, 1>",
                                "k_fwd_cols<8, 4, 6>", "k_fwd_rows<8, 6>", "k_inv_a<8, false, false, true>", "k_inv_b<8, false, true>"],
}
# known spills of the shipped build in kernels the workloads DO launch (none tolerated silently: list them here with the reason)
TOLERATED = {
    "k_fwd_rows<10, 4>": 16,         # cfg1 (the reference's CPU-sized case): 12 bytes since round 3, measured irrelevant there
    "k_fwd_cols_dual<1>": 24,        # cfg1opt pass 1: 20 bytes since round 3
    "k_conv1<13>": 20,               # after8k: one part per 2^14-point tile, four radix stages per transform (r05 experiments 10); 12 bytes with
                                     #   non-temporal loads, 20 with plain ones -- and 2914 -> 2745 us per block (experiments 11 g)
    "k_conv3_b<20>": 28,             # n_fft = 2^20: 4096-point rows, three full radix-16 stages per transform; still 2.8x the four passes
    "k_fb_plain<7, 0>": 32,          # raw words of the less common input forms parked during the decode (228 VGPRs; the sixteen 64-bit
    "k_fb_plain<10, 0>": 32,         #   sample indices as arrays had cost 48-64 bytes and 7 % of the kernel: r05 experiments 8)
    "k_fb_plain<7, 2>": 16,
}


@pytest.mark.parametrize("workload", sorted(HOT))
def test_hot_kernels_use_no_scratch(kernels, workload):
    for name in HOT[workload]:
        assert name in kernels, "%s: kernel %s not in the library (renamed? update this list)" % (workload, name)
        kd = kernels[name]
        scratch = int(kd.get(".private_segment_fixed_size", 0))
        assert scratch <= TOLERATED.get(name, 0), "%s: %s uses %d bytes of scratch per lane (%d VGPRs): a register spill in a hot kernel" % (
            workload, name, scratch, kd.get(".vgpr_count", -1))
        assert int(kd.get(".vgpr_count", 0)) <= 256
