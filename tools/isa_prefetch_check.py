"""Scan `hipcc -S` listings for load bursts that are waited for at once -- a prefetch that overlaps nothing.

    python tools/isa_prefetch_check.py /tmp/fb1.s [name-filter]

For every kernel: each run of >= 8 global loads (no barrier in between) and, behind it, the number of instructions up to the
first `s_waitcnt vmcnt(N)` with N < half the burst, and whether a barrier or LDS instruction comes first.  A burst whose wait
follows within a few dozen non-LDS instructions is the pattern k_inv_a had (round 4: copies of a conditional prefetch into
its loop-carried registers, placed by the compiler straight behind the loads)."""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name, body, kernels = None, [], []
for l in txt:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        name, body = m.group(1), []
        continue
    if name is not None:
        body.append(l)
        if "s_endpgm" in l:
            kernels.append((name, body)); name = None
for name, body in kernels:
    if flt and flt not in name:
        continue
    ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
    i, out = 0, []
    while i < len(ins):
        if ins[i].startswith("global_load"):
            j, n, last = i, 0, i
            while j < len(ins) and j - last < 12 and not ins[j].startswith(("s_barrier", "s_cbranch", "s_branch")):
                if ins[j].startswith("global_load"):
                    n += 1; last = j
                j += 1
            if n >= 8:
                k, what = last + 1, None
                while k < len(ins) and k - last < 400:
                    m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ins[k])
                    if m and int(m.group(1)) < n // 2:
                        what = "vmcnt(%s) after %d instr" % (m.group(1), k - last); break
                    if ins[k].startswith("s_barrier"):
                        what = "barrier first (%d instr)" % (k - last); break
                    k += 1
                out.append("  burst of %2d loads at %5d: %s" % (n, i, what))
            i = last + 1
        else:
            i += 1
    if out:
        short = re.sub(r"^_ZN9dspsr_amd\d+", "", name)[:60]
        print(short, "(%d instr)" % len(ins)); print("\n".join(out))
