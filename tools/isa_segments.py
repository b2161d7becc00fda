"""Per-segment instruction mix of a kernel: segments are delimited by s_barrier.  usage: isa_segments.py file.s NAME_SUBSTRING"""
import collections, sys
sys.path.insert(0, __file__.rsplit('/', 1)[0])
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith('s_endpgm'))
def cls(i):
    if i.startswith('v_pk_'): return 'pk'
    if i.startswith(('ds_',)): return 'lds'
    if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'gmem'
    if i.startswith('s_'): return 'salu'
    if i.startswith(('v_mov', 'v_accvgpr', 'v_swap')): return 'mov'
    if i.startswith(('v_cvt', 'v_sin', 'v_cos')): return 'cvt'
    if i.startswith(('v_mul_f32', 'v_fma', 'v_add_f32', 'v_sub_f32', 'v_fmac', 'v_mac')): return 'fp'
    if i.startswith('v_'): return 'int'
    return 'oth'
seg, segs, labels = collections.Counter(), [], []
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(('.', ';')):
        continue
    if l.endswith(':'):
        seg['label:' + l] += 0
        continue
    i = l.split()[0]
    if i.startswith('s_barrier'):
        segs.append(seg); seg = collections.Counter()
        continue
    seg[cls(i)] += 1
segs.append(seg)
print('%3s %6s %6s %6s %6s %6s %6s %6s %6s' % ('seg', 'pk', 'fp', 'cvt', 'int', 'mov', 'lds', 'gmem', 'salu'))
for n, s in enumerate(segs):
    print('%3d %6d %6d %6d %6d %6d %6d %6d %6d   %s' % (n, s['pk'], s['fp'], s['cvt'], s['int'], s['mov'], s['lds'], s['gmem'], s['salu'],
          ' '.join(k[6:] for k in s if k.startswith('label:'))[:60]))
