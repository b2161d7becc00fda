"""Instruction mix of one kernel in a device assembly listing (hipcc -S --cuda-device-only ...).
usage: python tools/isa_mix.py file.s SUBSTRING_OF_MANGLED_NAME"""
import collections
import re
import sys

INT = ('v_add_u', 'v_lshl', 'v_and', 'v_or', 'v_xor', 'v_lshr', 'v_mad_u', 'v_mul_u', 'v_mul_lo', 'v_mul_hi', 'v_sub_u', 'v_bfe',
       'v_add3', 'v_add_co', 'v_addc', 'v_ashr', 'v_sub_co', 'v_perm', 'v_cndmask', 'v_cmp', 'v_bfi', 'v_mad_i', 'v_sub_nc', 'v_add_nc',
       'v_mad_co', 'v_subrev', 'v_alignbit', 'v_readlane', 'v_readfirstlane', 'v_writelane', 'v_not', 'v_min_u', 'v_max_u', 'v_mbcnt',
       'v_subb', 'v_lshrrev', 'v_lshlrev', 'v_ashrrev')
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith('s_endpgm'))
c = collections.Counter()
n = 0
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    i = l.split()[0]
    n += 1
    if i.startswith('v_pk_'):
        k = 'v_pk (fp32 x2)'
    elif i.startswith(('ds_', 'buffer_', 'global_', 'flat_', 'scratch_')):
        k = i.split('_')[0] + ('_load' if ('load' in i or 'read' in i) else '_store')
    elif i.startswith('s_'):
        k = 's_waitcnt' if 'waitcnt' in i else 's_barrier' if 'barrier' in i else 's_other'
    elif i.startswith(('v_mov', 'v_accvgpr', 'v_swap')):
        k = 'v_mov'
    elif i.startswith(INT):
        k = 'v_int'
    elif i.startswith(('v_cvt', 'v_sin', 'v_cos', 'v_rcp', 'v_sqrt')):
        k = 'v_cvt/trans'
    elif i.startswith('v_'):
        k = 'v_fp32 scalar'
    else:
        k = 'other'
    c[k] += 1
print(lines[start].split(':')[0][:90], 'instructions', n)
for k, v in c.most_common():
    print('   %-16s %5d  %4.1f %%' % (k, v, 100.0 * v / n))
